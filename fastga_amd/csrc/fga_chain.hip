// fga_chain.hip -- diagonal-band chain detection over the sorted seed records, on the device.
//
// Replaces the chain scan of align_contigs (reference FastGA.c:3016-3176, 3340-3403); same semantics as the host
// routine fga_chain_scan (fga_chain.c, kept for keys that already live on the host), but the 16 B/record key stream
// never leaves HBM -- only the few thousand hits do.
//
// A *unit* is a bucket run d of one (strand, A contig, B contig) segment plus the directly following run when that
// is bucket d+1 (aux); units are independent of one another.  For one unit the reference's sequential scan has a
// closed form over the records r_0.. merged by anti-diagonal (ties: bucket d first):
//     cps_i = anti_i + 2 lcp_i          M_i = max(-CHAIN_BREAK, max_{j<=i} cps_j)         (ahgh is a prefix max)
//     brk_i = anti_i >= M_{i-1} + CHAIN_BREAK                                              (record i starts a chain)
//     c_i   = max(0, cps_i - max(M_{i-1}, anti_i))                                         (coverage it adds)
//   and per chain [s,t]: cov = sum c_i, mix = OR of the run bits, dgmin/dgmax = min/max diag&63 (+64 for run d+1),
//   alow = anti_s, ahgh = M_t; the chain is a hit when cov >= CHAIN_MIN and (mix != 1 or the unit is "new").
//
// Kernels:
//   chain_small_kernel    one thread per record; a thread sitting on a bucket head walks its unit (<= SMALL_LIMIT
//                         records, the overwhelming majority: noise buckets of one or two seeds), drops it when
//                         sum 2 lcp < CHAIN_MIN (no chain can reach the threshold) and otherwise runs the
//                         sequential scan; longer units are queued.
//   chain_plan_kernel     run ends and segment ranges of the queued units.
//   chain_segment_kernel  persistent wavefronts, one 4096-record segment of a long unit at a time: 64-record tiles
//                         of the two-run merge (ranks by binary search in LDS), wave prefix max for M, segmented
//                         wave scans for the chain statistics.
//   chain_stitch_kernel   reassembles the chains that cross segment boundaries, hits appended in order.
// Hits of a unit are contiguous (first_hit, nhits); units come back in arbitrary order and are sorted by the index
// of their first record on the host, which is the reference's order.
#include "fga_device.hpp"

typedef unsigned __int128 u128;

#define BUCK_SHIFT  6
#define BUCK_WIDTH  64
#define SMALL_LIMIT 48          // units with more records go to the wave-parallel kernel
#define HIT_STAGE   128         // hits staged in LDS per big unit before the contiguous block is reserved

#define CTR_UNITS   32
#define CTR_WORDS   40
struct chain_args
  { const uint4 *keys; int64_t n;
    int s_buck, s_b, s_a, s_strand, wd, wt, wb, wa;
    int64_t cbreak, cmin, amxpos, bmxpos;
    const int64_t *alen;
    fga_hit  *hits;   int64_t hit_cap;
    fga_unit *units;  int64_t *unit_head; int64_t unit_cap;
    int64_t  *bigq;   int64_t big_cap;
    unsigned long long *ctr;        // [0] hits, [CTR_UNITS] units (a cache line of its own: the two are the hot ones),
                                    // [2] queued long units, [3] segment cursor, [4] segments,
                                    // [5] staged inner hits
    int small_limit;
  };

__device__ __forceinline__ u128 key_at(const chain_args &G, int64_t i)
{ const uint4 k = G.keys[i];
  return ((u128) (((uint64_t) k.w << 32) | k.z) << 64) | (((uint64_t) k.y << 32) | k.x);
}
__device__ __forceinline__ int64_t key_anti(const chain_args &G, u128 k)
{ return (int64_t) ((uint64_t) (k >> 12) & ((1ull << G.wt) - 1)); }
__device__ __forceinline__ int key_lcp(u128 k)  { return (int) ((uint32_t) k & 63); }
__device__ __forceinline__ int key_drem(u128 k) { return (int) (((uint32_t) k >> 6) & 63); }

// chain_small_kernel's view of the key stream: the workgroup's 256 records, the one before them and the SMALL_LOOK after
// them sit in LDS (one coalesced pass); a unit of at most SMALL_LIMIT records that starts in the workgroup's range lies
// inside, so the walks of the bucket heads -- chains of dependent reads, one record at a time -- never wait for HBM.
#define SMALL_LOOK  64          // > SMALL_LIMIT + 1
struct key_tile
  { const uint4 *lds;           // record lo + t at lds[t]
    int64_t      lo, hi;        // records [lo,hi) are in the tile
  };

__device__ __forceinline__ u128 key_at(const chain_args &G, const key_tile &K, int64_t i)
{ const uint4 k = (i >= K.lo && i < K.hi) ? K.lds[i-K.lo] : G.keys[i];
  return ((u128) (((uint64_t) k.w << 32) | k.z) << 64) | (((uint64_t) k.y << 32) | k.x);
}

struct unit_info
  { int64_t b, m, e;            // run d = [b,m), run d+1 = [m,e)
    int     isnew, aux, comp, actg, bctg;
    int64_t cdiag, doffset, aoffset;
  };

__device__ __forceinline__ void unit_coords(const chain_args &G, u128 kb, unit_info &U)
{ U.comp = (int) ((uint64_t) (kb >> G.s_strand) & 1);
  U.actg = (int) ((uint64_t) (kb >> G.s_a) & ((1ull << G.wa) - 1));
  U.bctg = (int) ((uint64_t) (kb >> G.s_b) & ((1ull << G.wb) - 1));
  U.cdiag = (int64_t) ((uint64_t) (kb >> G.s_buck) & ((1ull << G.wd) - 1));
  const int64_t alen = G.alen[U.actg];
  U.doffset = alen - (G.amxpos + G.bmxpos);
  U.aoffset = alen - G.amxpos;
}

__device__ __forceinline__ fga_hit make_hit(const chain_args &G, const unit_info &U, int dgmin, int dgmax,
                                            int64_t alow, int64_t ahgh, int64_t cov)
{ fga_hit H;
  int64_t gmin = dgmin + (U.cdiag << BUCK_SHIFT), gmax = dgmax + (U.cdiag << BUCK_SHIFT);
  if (U.comp)
    { gmin += U.doffset; gmax += U.doffset;
      alow += U.aoffset; ahgh += U.aoffset;
    }
  else
    { gmin -= G.bmxpos; gmax -= G.bmxpos; }
  H.dgmin = (int32_t) gmin; H.dgmax = (int32_t) gmax;
  H.alow = alow; H.ahgh = ahgh;
  H.cov = (int32_t) cov; H.pad = 0;
  return H;
}

// The reference's sequential scan of one unit (FastGA.c:3060-3176 as restated in fga_chain.c); WRITE = false only
// counts the hits.  (Keeping the first two hits of the counting pass in registers, so that almost no unit is scanned
// twice, made the kernel slower: 10.1 -> 12.2 ms.)
template <bool WRITE>
__device__ int scan_sequential(const chain_args &G, const key_tile &K, const unit_info &U, fga_hit *out)
{ const int64_t CB = G.cbreak, CMIN = G.cmin;
  int64_t s = U.b, t = U.m;
  u128 ks = key_at(G,K,s), kt = 0;
  int64_t ipost = key_anti(G,ks), apost = INT64_MAX;
  if (U.aux)
    { kt = key_at(G,K,t); apost = key_anti(G,kt); }
  int64_t ahgh = -CB, alow = (apost < ipost) ? apost : ipost, anti, cov = 0;
  int dgmin = 2*BUCK_WIDTH, dgmax = 0, dg, lcp, wch, mix = 0, go = 1, nh = 0;
  while (go)
    { if (apost < ipost)
        { lcp = key_lcp(kt); dg = key_drem(kt) + BUCK_WIDTH; anti = apost;
          t += 1;
          if (t >= U.e) apost = INT64_MAX;
          else { kt = key_at(G,K,t); apost = key_anti(G,kt); }
          wch = 0x2;
        }
      else
        { anti = ipost;
          if (s < U.m) { lcp = key_lcp(ks); dg = key_drem(ks); }
          else         lcp = dg = 0;
          s += 1;
          if (s >= U.m)
            { if (s > U.m) go = 0;
              else         ipost = INT64_MAX;
            }
          else
            { ks = key_at(G,K,s); ipost = key_anti(G,ks); }
          wch = 0x1;
        }
      lcp <<= 1;
      if (anti < ahgh + CB)
        { const int64_t cps = anti + lcp;
          if (cps > ahgh)
            { cov += (anti >= ahgh) ? lcp : cps-ahgh;
              ahgh = cps;
            }
          mix |= wch;
          if (dg < dgmin) dgmin = dg;
          else if (dg > dgmax) dgmax = dg;
        }
      else
        { if (cov >= CMIN && (mix != 1 || U.isnew))
            { if (WRITE)
                out[nh] = make_hit(G,U,dgmin,dgmax,alow,ahgh,cov);
              nh += 1;
            }
          if (go)
            { cov = lcp; ahgh = anti + lcp; mix = wch; alow = anti; dgmin = dgmax = dg; }
        }
    }
  return nh;
}

__device__ __forceinline__ void store_unit(const chain_args &G, const unit_info &U, int64_t first, int nh, unsigned long long u)
{ if ((int64_t) u < G.unit_cap)
    { fga_unit R;
      R.actg = U.actg; R.bctg = U.bctg; R.comp = U.comp; R.nhits = nh;
      R.first_hit = first; R.bucket = U.cdiag;
      G.units[u] = R;
      G.unit_head[u] = U.b;
    }
}

__device__ __forceinline__ void publish_unit(const chain_args &G, const unit_info &U, int64_t first, int nh)
{ store_unit(G,U,first,nh,atomicAdd(G.ctr+CTR_UNITS,1ull)); }

// One thread per record.  What the kernel is made of was found by taking it apart (round 4): the hit and unit slots of a
// WORKGROUP come from one atomic each (per unit, and again per wavefront, the kernel took as long as its atomics: the two
// counters share a cache line and a line serves ~88 atomics per microsecond); a bucket head does not WALK its unit -- the
// walk is a loop of 128-bit shifts and compares whose longest instance in a wavefront (49 steps, twice) every lane waits
// for -- but reads the unit's end off a bit mask of the bucket heads of the workgroup's records and the bound of its
// coverage off a prefix sum of their lcp values; and the sequential scans of the units that can reach the threshold run
// compacted, one unit per lane of full wavefronts.  22 ms -> 9 ms for the 1.01 M units of the 150 Mbp self comparison
// (workgroups of 256 / 512 / 1024 records: 14 / 9 / 12.5 ms -- fewer atomics against fewer workgroups per CU to overlap the
// phases between the barriers); on the bench pair, where almost no bucket can reach the threshold, 0.4 -> 0.7 ms.

// first bucket head after offset o of the workgroup's range (> o), as far as two mask words reach; else a large number
template <int SMALL_WORDS>
__device__ __forceinline__ int next_head(const uint64_t *mask, int o)
{ const int w = (o+1) >> 6, b = (o+1) & 63;
  uint64_t x = mask[w] >> b;
  if (x != 0)
    return o + 1 + (__ffsll((long long) x) - 1);
  if (w+1 < SMALL_WORDS && (x = mask[w+1]) != 0)
    return (w+1)*64 + (__ffsll((long long) x) - 1);
  return 1 << 20;
}

template <int SMALL_BLOCK>
__global__ __launch_bounds__(SMALL_BLOCK)
void chain_small_kernel(chain_args G)
{ constexpr int SMALL_WORDS = SMALL_BLOCK/64 + 1;   // head-mask words: the block's records and SMALL_LOOK (= 64) beyond
  __shared__ uint4 tile[SMALL_BLOCK + 1 + SMALL_LOOK];
  __shared__ int wave_hits[SMALL_BLOCK/64 + 1], wave_units[SMALL_BLOCK/64];
  __shared__ unsigned long long block_base[2];
  __shared__ uint32_t list[SMALL_BLOCK];
  __shared__ uint64_t heads[SMALL_WORDS];
  __shared__ int      lsum[SMALL_BLOCK + SMALL_LOOK + 1];       // lsum[o] = sum of the lcp values of offsets < o
  const int64_t base = (int64_t) blockIdx.x * blockDim.x;
  const int64_t i = base + threadIdx.x;
  const int lane = (int) (threadIdx.x & 63);
  const int wave = (int) (threadIdx.x >> 6);
  key_tile K;
  K.lds = tile;
  K.lo = base - 1;
  K.hi = K.lo + SMALL_BLOCK + 1 + SMALL_LOOK;
  if (K.hi > G.n) K.hi = G.n;
  for (int64_t t = K.lo + threadIdx.x; t < K.hi; t += SMALL_BLOCK)
    if (t >= 0)
      tile[t-K.lo] = G.keys[t];
  if (K.lo < 0) K.lo = 0, K.lds = tile + 1;
  __syncthreads();

  // bucket heads and lcp values of the offsets 0 .. SMALL_BLOCK+SMALL_LOOK-1 (position n counts as a head: the end)
  const uint64_t dmask = (1ull << G.wd) - 1;
  u128 Ui = 0;
  bool head = false, prevadj = false;
  for (int r = 0; r < 2; r++)
    { const int o = r*SMALL_BLOCK + (int) threadIdx.x;
      if (r == 1 && o >= SMALL_BLOCK + SMALL_LOOK)
        break;                                               // whole wavefronts leave: offsets come in 64s
      const int64_t p = base + o;
      bool h = (p == G.n);
      int  l = 0;
      if (p < G.n)
        { const u128 k = key_at(G,K,p);
          const u128 Uk = k >> G.s_buck;
          bool adj = false;
          h = true;
          l = key_lcp(k);
          if (p > 0)
            { const u128 Up = key_at(G,K,p-1) >> G.s_buck;
              h = (Up != Uk);
              adj = (Up + 1 == Uk) && (((uint64_t) Uk & dmask) != 0);
            }
          if (r == 0)
            { Ui = Uk; head = h; prevadj = adj; }
        }
      const uint64_t hm = __ballot(h);
      int incl = l;
      #pragma unroll
      for (int d = 1; d < 64; d <<= 1)
        { const int t = __shfl_up(incl,d,64);
          if (lane >= d) incl += t;
        }
      const int wo = o >> 6;
      if (lane == 0)
        heads[wo] = hm;
      if (lane == 63)
        wave_hits[wo] = incl;                               // the wavefront's lcp total (the array is free until the slots)
      lsum[o+1] = incl;                                       // completed below with the wavefronts before
    }
  __syncthreads();
  { int before = 0;                                           // lcp total of the wavefronts before this thread's offsets
    for (int w = 0; w < wave; w++)
      before += wave_hits[w];
    int all = before;
    for (int w = wave; w < SMALL_BLOCK/64; w++)
      all += wave_hits[w];
    __syncthreads();
    lsum[threadIdx.x+1] += before;
    if (threadIdx.x < SMALL_LOOK)
      lsum[SMALL_BLOCK+threadIdx.x+1] += all;
    if (threadIdx.x == 0)
      lsum[0] = 0;
  }
  __syncthreads();

  unit_info U;
  int nh = 0;
  bool live = (i < G.n) && head;
  if (live)
    { const int o = (int) threadIdx.x, lim = G.small_limit;
      int m = next_head<SMALL_WORDS>(heads,o), e;
      U.b = i; U.isnew = !prevadj;
      if (m-o > lim)
        e = m;                                                // a long unit
      else
        { e = m;
          if (((uint64_t) Ui & dmask) != dmask && base+m < G.n && (key_at(G,K,base+m) >> G.s_buck) == Ui + 1)
            e = next_head<SMALL_WORDS>(heads,m);
        }
      if (e-o > lim)                                          // a long unit: one wavefront will take it
        { const unsigned long long q = atomicAdd(G.ctr+2,1ull);
          if ((int64_t) q < G.big_cap)
            G.bigq[q] = i;
          live = false;
        }
      else
        { const int bound = 2*(lsum[e]-lsum[o]);              // sum of 2 lcp over the unit: an upper bound of any chain's cov
          U.m = base+m; U.e = base+e; U.aux = (e > m);
          if ((!U.isnew && !U.aux) || bound < G.cmin)
            live = false;
        }
    }
  // the units that can reach the threshold, compacted over the workgroup
  { const uint64_t lm = __ballot(live);
    if (lane == 0)
      wave_units[wave] = __popcll(lm);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < SMALL_BLOCK/64; w++)
      { const int c = wave_units[w];
        if (w < wave) base += c;
        total += c;
      }
    if (live)
      list[base + __popcll(lm & ((1ull << lane) - 1))] = (uint32_t) threadIdx.x | ((uint32_t) (U.m-U.b) << 10) |
                                                         ((uint32_t) (U.e-U.b) << 16) | ((uint32_t) U.isnew << 22);
    __syncthreads();
    live = (int) threadIdx.x < total;
  }
  if (live)
    { const uint32_t d = list[threadIdx.x];
      U.b = base + (d & 1023);
      U.m = U.b + ((d >> 10) & 63); U.e = U.b + ((d >> 16) & 63);
      U.aux = (U.e > U.m); U.isnew = (int) ((d >> 22) & 1);
      unit_coords(G,key_at(G,K,U.b),U);
      nh = scan_sequential<false>(G,K,U,NULL);
    }
  // slots for the workgroup's units and hits (every thread is here)
  const uint64_t em = __ballot(nh > 0);
  int incl = nh;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { const int t = __shfl_up(incl,d,64);
      if (lane >= d) incl += t;
    }
  if (lane == 63)
    { wave_hits[wave] = incl; wave_units[wave] = __popcll(em); }
  __syncthreads();
  if (threadIdx.x == 0)
    { int h = 0, u = 0;
      for (int w = 0; w < SMALL_BLOCK/64; w++)
        { const int hw = wave_hits[w], uw = wave_units[w];
          wave_hits[w] = h; wave_units[w] = u;
          h += hw; u += uw;
        }
      if (u > 0)
        { block_base[0] = atomicAdd(G.ctr+0,(unsigned long long) h);
          block_base[1] = atomicAdd(G.ctr+CTR_UNITS,(unsigned long long) u);
        }
    }
  __syncthreads();
  if (nh == 0)
    return;
  const int64_t first = (int64_t) block_base[0] + wave_hits[wave] + (incl - nh);
  if (first + nh <= G.hit_cap)
    scan_sequential<true>(G,K,U,G.hits + first);
  store_unit(G,U,first,nh,block_base[1] + (unsigned long long) (wave_units[wave] + __popcll(em & ((1ull << lane) - 1))));
}

// The same result for key streams whose buckets are noise -- one unit in 10^4 records can reach the threshold (the bench
// pair: 2,087 units in 48.6 M records): one thread per record straight from HBM, a bucket head walks the one or two
// records of its unit, slots by one pair of atomics per wavefront that has a unit at all.  0.38 ms on the bench pair
// against the 0.7 ms of the workgroup-tiled kernel above, 22 against 9 ms where units are dense: fga_chain_scan_device
// picks by the unit density of the device's previous launch.
__global__ __launch_bounds__(256)
void chain_sparse_kernel(chain_args G)
{ key_tile K;
  K.lds = NULL; K.lo = K.hi = 0;                     // nothing staged: every key from HBM
 const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int) (threadIdx.x & 63);
  unit_info U;
  int nh = 0;
  bool live = i < G.n;
  u128 ki = 0;
  if (live)
    { ki = key_at(G,i);
      const u128 Ui = ki >> G.s_buck;
      const uint64_t dmask = (1ull << G.wd) - 1;
      bool head = true, prevadj = false;
      if (i > 0)
        { const u128 Up = key_at(G,i-1) >> G.s_buck;
          head = (Up != Ui);
          prevadj = (Up + 1 == Ui) && (((uint64_t) Ui & dmask) != 0);
        }
      live = head;
      if (live)
        { U.b = i; U.isnew = !prevadj;
          const int64_t lim = G.small_limit;
          int64_t bound = 2*key_lcp(ki);                   // sum of 2 lcp over the unit: an upper bound of any chain's cov
          int64_t m = i+1;
          while (m < G.n && m-i <= lim)
            { const u128 k = key_at(G,m);
              if ((k >> G.s_buck) != Ui) break;
              bound += 2*key_lcp(k);
              m += 1;
            }
          int64_t e = m;
          if (m-i <= lim && ((uint64_t) Ui & dmask) != dmask)
            { const u128 Un = Ui + 1;
              while (e < G.n && e-i <= lim)
                { const u128 k = key_at(G,e);
                  if ((k >> G.s_buck) != Un) break;
                  bound += 2*key_lcp(k);
                  e += 1;
                }
            }
          if (e-i > lim)                                   // a long unit: one wavefront will take it
            { const unsigned long long q = atomicAdd(G.ctr+2,1ull);
              if ((int64_t) q < G.big_cap)
                G.bigq[q] = i;
              live = false;
            }
          else
            { U.m = m; U.e = e; U.aux = (e > m);
              if ((!U.isnew && !U.aux) || bound < G.cmin)
                live = false;
            }
        }
    }
  if (live)
    { unit_coords(G,ki,U);
      nh = scan_sequential<false>(G,K,U,NULL);
    }
  // slots for the wavefront's units and hits (every lane is here)
  const uint64_t em = __ballot(nh > 0);
  if (em == 0)
    return;
  int incl = nh;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { const int t = __shfl_up(incl,d,64);
      if (lane >= d) incl += t;
    }
  const int total = __shfl(incl,63,64);
  unsigned long long hbase = 0, ubase = 0;
  if (lane == 0)
    { hbase = atomicAdd(G.ctr+0,(unsigned long long) total);
      ubase = atomicAdd(G.ctr+CTR_UNITS,(unsigned long long) __popcll(em));
    }
  hbase = ((unsigned long long) (uint32_t) __shfl((int) (uint32_t) (hbase >> 32),0,64) << 32) | (uint32_t) __shfl((int) (uint32_t) hbase,0,64);
  ubase = ((unsigned long long) (uint32_t) __shfl((int) (uint32_t) (ubase >> 32),0,64) << 32) | (uint32_t) __shfl((int) (uint32_t) ubase,0,64);
  if (nh == 0)
    return;
  const int64_t first = (int64_t) hbase + (incl - nh);
  if (first + nh <= G.hit_cap)
    scan_sequential<true>(G,K,U,G.hits + first);
  store_unit(G,U,first,nh,ubase + (unsigned long long) __popcll(em & ((1ull << lane) - 1)));
}

// ---------------------------------------------------------------------------------------------------
// long units: split into segments of SEG_RECS merged records, one wavefront per segment, then stitched
// ---------------------------------------------------------------------------------------------------
// A contig-long alignment puts hundreds of thousands of records into one unit, far too many for one wavefront.
// The closed form above makes segments of the merged sequence nearly independent:
//   * M at a segment start is the max of cps over earlier records; cps_j <= anti_j + 2*63 and anti is sorted, so
//     only records within 126 of the boundary's anti-diagonal matter: a short look-back, no chain of carries;
//   * chains that cross segment boundaries are reassembled from per-segment partials: the records before the
//     segment's first chain start (head), the chains wholly inside (inner hits, in order) and the records after
//     its last chain start (tail).  The stitch is sequential but only touches one record per segment.
#define SEG_RECS  4096
#define LCP2_MAX  126

struct big_unit
  { int64_t b, m, e;
    int32_t isnew, aux;
    int64_t seg_base; int32_t nseg, pad;
  };

struct seg_out
  { int64_t head_cov, head_ahgh;            // records before the first chain start; M just before that start
    int32_t head_mix, head_dmin, head_dmax, has_brk;
    int64_t tail_cov, tail_alow, m_out;     // the chain open at the end of the segment; M at its last record
    int32_t tail_mix, tail_dmin, tail_dmax, ihit_n;
    int64_t ihit_first;                     // inner hits: stage[ihit_first .. +ihit_n)
  };

struct big_args
  { big_unit *bunits;
    int32_t  *segmap;      int64_t seg_cap;
    seg_out  *segs;
    fga_hit  *stage;       int64_t stage_cap;       // inner hits of all segments (ctr[5] counts them)
  };

struct big_shared
  { int64_t  sA[64], sB[64];          // anti-diagonals of the two 64-record candidate windows (INT64_MAX: none)
    uint64_t merged[64];              // the next 64 records of the merge: anti << 20 | run bit << 16 | dg << 8 | lcp
    fga_hit  stage[HIT_STAGE];
  };

__device__ __forceinline__ int64_t shfl_up64(int64_t v, int d)
{ const int lo = __shfl_up((int) (uint32_t) (uint64_t) v,d,64), hi = __shfl_up((int) (uint32_t) ((uint64_t) v >> 32),d,64);
  return (int64_t) (((uint64_t) (uint32_t) hi << 32) | (uint32_t) lo);
}
__device__ __forceinline__ int64_t bcast64(int64_t v, int l)
{ const int lo = __shfl((int) (uint32_t) (uint64_t) v,l,64), hi = __shfl((int) (uint32_t) ((uint64_t) v >> 32),l,64);
  return (int64_t) (((uint64_t) (uint32_t) hi << 32) | (uint32_t) lo);
}
__device__ __forceinline__ int64_t wave_max64(int64_t v)
{ for (int d = 32; d > 0; d >>= 1)
    { const int lo = __shfl_xor((int) (uint32_t) (uint64_t) v,d,64), hi = __shfl_xor((int) (uint32_t) ((uint64_t) v >> 32),d,64);
      const int64_t t = (int64_t) (((uint64_t) (uint32_t) hi << 32) | (uint32_t) lo);
      if (t > v) v = t;
    }
  return v;
}

// first index in [lo,hi) whose bucket id (key >> s_buck) exceeds U
__device__ int64_t upper_bound_bucket(const chain_args &G, int64_t lo, int64_t hi, u128 U)
{ int64_t step = 64;
  int64_t p = lo;                                   // gallop first: runs are short compared with the array
  while (p < hi)
    { int64_t q = p + step < hi ? p + step : hi;
      if ((key_at(G,q-1) >> G.s_buck) > U) { hi = q; break; }
      p = q;
      step <<= 1;
    }
  lo = p;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      if ((key_at(G,mid) >> G.s_buck) > U) hi = mid;
      else lo = mid + 1;
    }
  return lo;
}

// plan: one thread per queued unit -- run ends, new/aux flags, segment range
__global__ __launch_bounds__(64)
void chain_plan_kernel(chain_args G, big_args B, int64_t nbig)
{ const int64_t q = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nbig)
    return;
  const uint64_t dmask = (1ull << G.wd) - 1;
  big_unit U;
  U.b = G.bigq[q];
  const u128 Ub = key_at(G,U.b) >> G.s_buck;
  U.isnew = 1;
  if (U.b > 0)
    { const u128 Up = key_at(G,U.b-1) >> G.s_buck;
      if ((Up + 1 == Ub) && (((uint64_t) Ub & dmask) != 0))
        U.isnew = 0;
    }
  U.m = upper_bound_bucket(G,U.b+1,G.n,Ub);
  U.e = U.m;
  if (((uint64_t) Ub & dmask) != dmask && U.m < G.n && (key_at(G,U.m) >> G.s_buck) == Ub + 1)
    U.e = upper_bound_bucket(G,U.m+1,G.n,Ub+1);
  U.aux = (U.e > U.m);
  U.nseg = 0; U.seg_base = 0; U.pad = 0;
  if (U.isnew || U.aux)
    { U.nseg = (int32_t) ((U.e - U.b + SEG_RECS - 1) / SEG_RECS);
      U.seg_base = (int64_t) atomicAdd(G.ctr+4,(unsigned long long) U.nseg);
      for (int s = 0; s < U.nseg; s++)
        if (U.seg_base + s < B.seg_cap)
          B.segmap[U.seg_base + s] = (int32_t) q;
    }
  B.bunits[q] = U;
}

// merge-path split: how many records of run d are among the first r of the merge (run d first on ties)
__device__ int64_t merge_split(const chain_args &G, const big_unit &U, int64_t r)
{ const int64_t na = U.m - U.b, nb = U.e - U.m;
  int64_t lo = r > nb ? r - nb : 0, hi = r < na ? r : na;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1, j = r - mid;
      // P(mid): A[mid] > B[j-1] (or a boundary); false -> A[mid] belongs to the first r as well
      bool P = true;
      if (mid < na && j > 0)
        P = key_anti(G,key_at(G,U.b+mid)) > key_anti(G,key_at(G,U.m+j-1));
      if (P) hi = mid; else lo = mid + 1;
    }
  return lo;
}

// M just before record (ia,ib) of the merge: the max of cps over the earlier records; only those within
// LCP2_MAX of the last earlier anti-diagonal can hold it
__device__ int64_t lookback_max(const chain_args &G, const big_unit &U, int64_t ia, int64_t ib)
{ const int lane = threadIdx.x & 63;
  if (ia == U.b && ib == U.m)
    return -G.cbreak;
  int64_t antiL = -1;
  if (ia > U.b) antiL = key_anti(G,key_at(G,ia-1));
  if (ib > U.m)
    { const int64_t t = key_anti(G,key_at(G,ib-1));
      if (t > antiL) antiL = t;
    }
  const int64_t floor_ = antiL - LCP2_MAX;
  int64_t best = INT64_MIN;
  for (int run = 0; run < 2; run++)
    { const int64_t first = run ? U.m : U.b;
      int64_t p = run ? ib : ia;                    // records [first,p) of this run precede the segment
      while (p > first)
        { const int64_t idx = p - 1 - lane;
          int64_t anti = INT64_MIN, cps = INT64_MIN;
          if (idx >= first)
            { const u128 k = key_at(G,idx);
              anti = key_anti(G,k);
              if (anti >= floor_)
                cps = anti + 2*key_lcp(k);
            }
          if (cps > best) best = cps;
          // the run is sorted: once a lane sees an anti below the floor (or the run start) nothing earlier counts
          if (__ballot(idx < first || anti < floor_))
            break;
          p -= 64;
        }
    }
  return wave_max64(best);
}

// One pass over one segment [ (ia0,ib0), (ia1,ib1) ) of a long unit.  Inner hits are written to `out` (global,
// second pass) or staged in LDS (first pass, the first HIT_STAGE of them); returns their number.
__device__ int scan_segment(const chain_args &G, const unit_info &U, big_shared *sh, int64_t ia0, int64_t ib0,
                            int64_t ia1, int64_t ib1, int64_t M_in, fga_hit *out, seg_out &SO)
{ const int lane = threadIdx.x & 63;
  const int64_t CB = G.cbreak, CMIN = G.cmin;
  int64_t ia = ia0, ib = ib0;
  int64_t M = M_in;
  // the records before the segment's first chain start continue whatever chain is open at the boundary: they are
  // accumulated into an (initially empty) carry that becomes the head partial
  bool    seen_brk = false;
  int64_t ccov = 0, calow = 0;
  int     cmix = 0, cdmin = 255, cdmax = 0;
  int     nh = 0;
  SO.head_cov = 0; SO.head_ahgh = M_in; SO.head_mix = 0; SO.head_dmin = 255; SO.head_dmax = 0;

#define EMIT(idx,H) { if (out != NULL) out[idx] = (H); else if ((idx) < HIT_STAGE) sh->stage[idx] = (H); }

  while (ia < ia1 || ib < ib1)
    { const int na = (int) ((ia1 - ia) < 64 ? (ia1 - ia) : 64), nb = (int) ((ib1 - ib) < 64 ? (ib1 - ib) : 64);
      const int total = (na + nb) < 64 ? (na + nb) : 64;
      int64_t antiA = INT64_MAX, antiB = INT64_MAX;
      uint32_t infoA = 0, infoB = 0;
      if (lane < na)
        { const u128 k = key_at(G,ia+lane);
          antiA = key_anti(G,k); infoA = (uint32_t) key_lcp(k) | ((uint32_t) key_drem(k) << 8);
        }
      if (lane < nb)
        { const u128 k = key_at(G,ib+lane);
          antiB = key_anti(G,k); infoB = (uint32_t) key_lcp(k) | ((uint32_t) (key_drem(k) + BUCK_WIDTH) << 8) | (1u << 16);
        }
      __syncthreads();
      sh->sA[lane] = antiA; sh->sB[lane] = antiB;
      __syncthreads();
      // rank in the merge: A before B on ties (the reference takes run d+1 only when apost < ipost)
      int rA = 64, rB = 64;
      if (lane < na)
        { int lo = 0, hi = 64;                        // number of B candidates < antiA
          while (lo < hi) { const int mid = (lo+hi) >> 1; if (sh->sB[mid] < antiA) lo = mid+1; else hi = mid; }
          rA = lane + lo;
        }
      if (lane < nb)
        { int lo = 0, hi = 64;                        // number of A candidates <= antiB
          while (lo < hi) { const int mid = (lo+hi) >> 1; if (sh->sA[mid] <= antiB) lo = mid+1; else hi = mid; }
          rB = lane + lo;
        }
      if (rA < 64) sh->merged[rA] = ((uint64_t) antiA << 20) | infoA;
      if (rB < 64) sh->merged[rB] = ((uint64_t) antiB << 20) | infoB;
      const int cntA = __popcll(__ballot(rA < 64)), cntB = __popcll(__ballot(rB < 64));
      __syncthreads();

      const bool valid = lane < total;
      const uint64_t rec = valid ? sh->merged[lane] : 0;
      const int64_t anti = (int64_t) (rec >> 20);
      const int lcp2 = 2 * (int) (rec & 63);
      const int dg  = (int) ((rec >> 8) & 255);
      const int wch = (rec & (1u << 16)) ? 2 : 1;
      const int64_t cps = anti + lcp2;

      // prefix max of cps -> M_{i-1} (exclusive) and M_i (inclusive), both including the carry
      int64_t x = valid ? cps : INT64_MIN;
      for (int d = 1; d < 64; d <<= 1)
        { const int64_t t = shfl_up64(x,d);
          if (lane >= d && t > x) x = t;
        }
      int64_t Mprev = shfl_up64(x,1);
      if (lane == 0 || Mprev < M) Mprev = M;
      const int64_t Minc = x > M ? x : M;
      const bool brk = valid && anti >= Mprev + CB;
      int64_t c = 0;
      if (valid)
        { const int64_t base = Mprev > anti ? Mprev : anti;
          c = cps > base ? cps - base : 0;
        }

      // the chain carried in from the previous tile ends here when lane 0 starts a new one
      const uint64_t brkmask = __ballot(brk);
      bool open = true;
      if (brkmask & 1)
        { if (!seen_brk)
            { SO.head_cov = ccov; SO.head_mix = cmix; SO.head_dmin = cdmin; SO.head_dmax = cdmax; SO.head_ahgh = M; }
          else if (ccov >= CMIN && (cmix != 1 || U.isnew))
            { if (lane == 0)
                EMIT(nh,make_hit(G,U,cdmin,cdmax,calow,M,ccov))
              nh += 1;
            }
          open = false;
        }

      // inclusive segmented scans, segment heads at brk; lanes before the first head continue the carried chain
      int64_t cov = c, alow = anti;
      int mix = valid ? wch : 0, dmin = valid ? dg : 255, dmax = valid ? dg : 0;
      int f = brk ? 1 : 0;
      for (int d = 1; d < 64; d <<= 1)
        { const int64_t tcov = shfl_up64(cov,d), talow = shfl_up64(alow,d);
          const int tmix = __shfl_up(mix,d,64), tmin = __shfl_up(dmin,d,64), tmax = __shfl_up(dmax,d,64);
          const int tf = __shfl_up(f,d,64);
          if (lane >= d)
            { if (!f)
                { cov += tcov; alow = talow; mix |= tmix;
                  if (tmin < dmin) dmin = tmin;
                  if (tmax > dmax) dmax = tmax;
                }
              f |= tf;
            }
        }
      if (!f && open)
        { cov += ccov; alow = calow; mix |= cmix;
          if (cdmin < dmin) dmin = cdmin;
          if (cdmax > dmax) dmax = cdmax;
        }

      // chain ends inside the tile: the next lane starts a new chain.  The one that has no start in this segment
      // (f == 0 before any start was seen) is the head partial, the others are complete chains.
      const bool nbrk = (brkmask >> (lane+1 < 64 ? lane+1 : 63)) & 1;
      const bool isend = valid && lane+1 < total && nbrk;
      const bool ishead = isend && !f && !seen_brk;
      const uint64_t headm = __ballot(ishead);
      if (headm)
        { const int l = __ffsll((unsigned long long) headm) - 1;
          SO.head_cov = bcast64(cov,l); SO.head_mix = __shfl(mix,l,64);
          SO.head_dmin = __shfl(dmin,l,64); SO.head_dmax = __shfl(dmax,l,64); SO.head_ahgh = bcast64(Minc,l);
        }
      const bool ishit = isend && !ishead && cov >= CMIN && (mix != 1 || U.isnew);
      const uint64_t hm = __ballot(ishit);
      if (ishit)
        { const int idx = nh + __popcll(hm & ((1ull << lane) - 1));
          EMIT(idx,make_hit(G,U,dmin,dmax,alow,Minc,cov))
        }
      nh += __popcll(hm);
      if (brkmask)
        seen_brk = true;

      // carry: the chain open at the last record of the tile
      const int last = total-1;
      ccov = bcast64(cov,last); calow = bcast64(alow,last);
      cmix = __shfl(mix,last,64); cdmin = __shfl(dmin,last,64); cdmax = __shfl(dmax,last,64);
      M = bcast64(Minc,last);
      ia += cntA; ib += cntB;
    }
  SO.has_brk = seen_brk ? 1 : 0;
  SO.m_out = M;
  if (!seen_brk)
    { SO.head_cov = ccov; SO.head_mix = cmix; SO.head_dmin = cdmin; SO.head_dmax = cdmax; SO.head_ahgh = M; }
  SO.tail_cov = ccov; SO.tail_alow = calow; SO.tail_mix = cmix; SO.tail_dmin = cdmin; SO.tail_dmax = cdmax;
  __syncthreads();
  return nh;
#undef EMIT
}

__global__ __launch_bounds__(64)
void chain_segment_kernel(chain_args G, big_args B, int64_t nsegs)
{ __shared__ big_shared sh;
  const int lane = threadIdx.x;
  while (1)
    { unsigned long long q = 0;
      if (lane == 0)
        q = atomicAdd(G.ctr+3,1ull);
      q = (unsigned long long) bcast64((int64_t) q,0);
      if ((int64_t) q >= nsegs)
        break;
      const big_unit BU = B.bunits[B.segmap[q]];
      const int64_t s = (int64_t) q - BU.seg_base;
      const int64_t nrec = BU.e - BU.b;
      const int64_t r0 = s * SEG_RECS, r1 = (r0 + SEG_RECS < nrec) ? r0 + SEG_RECS : nrec;
      const int64_t a0 = merge_split(G,BU,r0), a1 = merge_split(G,BU,r1);
      const int64_t ia0 = BU.b + a0, ib0 = BU.m + (r0 - a0), ia1 = BU.b + a1, ib1 = BU.m + (r1 - a1);
      unit_info U;
      U.b = BU.b; U.m = BU.m; U.e = BU.e; U.isnew = BU.isnew; U.aux = BU.aux;
      unit_coords(G,key_at(G,BU.b),U);
      const int64_t M_in = lookback_max(G,BU,ia0,ib0);
      seg_out SO;
      const int nh = scan_segment(G,U,&sh,ia0,ib0,ia1,ib1,M_in,NULL,SO);
      SO.ihit_n = nh; SO.ihit_first = 0;
      if (nh > 0)
        { unsigned long long first = 0;
          if (lane == 0)
            first = atomicAdd(G.ctr+5,(unsigned long long) nh);
          first = (unsigned long long) bcast64((int64_t) first,0);
          SO.ihit_first = (int64_t) first;
          if ((int64_t) first + nh <= B.stage_cap)
            { if (nh <= HIT_STAGE)
                { for (int q2 = lane; q2 < nh; q2 += 64)
                    B.stage[first + q2] = sh.stage[q2];
                }
              else
                { seg_out S2;
                  scan_segment(G,U,&sh,ia0,ib0,ia1,ib1,M_in,B.stage + first,S2);
                }
            }
        }
      if (lane == 0)
        B.segs[q] = SO;
      __syncthreads();
    }
}

// stitch: one thread per long unit walks its segments in order, closes the chains that cross segment
// boundaries and lays the unit's hits out contiguously
template <bool WRITE>
__device__ int stitch_unit(const chain_args &G, const big_args &B, const big_unit &BU, const unit_info &U, fga_hit *out)
{ const int64_t CMIN = G.cmin;
  int64_t cov = 0, alow = 0, M = -G.cbreak;
  int mix = 0, dmin = 255, dmax = 0, nh = 0;
  for (int s = 0; s < BU.nseg; s++)
    { const seg_out &S = B.segs[BU.seg_base + s];
      cov += S.head_cov; mix |= S.head_mix;
      if (S.head_dmin < dmin) dmin = S.head_dmin;
      if (S.head_dmax > dmax) dmax = S.head_dmax;
      if (S.has_brk)
        { if (cov >= CMIN && (mix != 1 || U.isnew))
            { if (WRITE) out[nh] = make_hit(G,U,dmin,dmax,alow,S.head_ahgh,cov);
              nh += 1;
            }
          if (WRITE)
            for (int q = 0; q < S.ihit_n; q++)
              out[nh+q] = B.stage[S.ihit_first + q];
          nh += S.ihit_n;
          cov = S.tail_cov; alow = S.tail_alow; mix = S.tail_mix; dmin = S.tail_dmin; dmax = S.tail_dmax;
        }
      M = S.m_out;
    }
  if (cov >= CMIN && (mix != 1 || U.isnew))                  // the reference's flush step
    { if (WRITE) out[nh] = make_hit(G,U,dmin,dmax,alow,M,cov);
      nh += 1;
    }
  return nh;
}

__global__ __launch_bounds__(64)
void chain_stitch_kernel(chain_args G, big_args B, int64_t nbig)
{ const int64_t q = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nbig)
    return;
  const big_unit BU = B.bunits[q];
  if (BU.nseg == 0)
    return;
  unit_info U;
  U.b = BU.b; U.m = BU.m; U.e = BU.e; U.isnew = BU.isnew; U.aux = BU.aux;
  unit_coords(G,key_at(G,BU.b),U);
  const int nh = stitch_unit<false>(G,B,BU,U,NULL);
  if (nh == 0)
    return;
  const int64_t first = (int64_t) atomicAdd(G.ctr+0,(unsigned long long) nh);
  if (first + nh <= G.hit_cap)
    stitch_unit<true>(G,B,BU,U,G.hits + first);
  publish_unit(G,U,first,nh);
}

// ---------------------------------------------------------------------------------------------------
// canonical order on the device: units by the index of their first record (the reference's order), hits re-laid in
// unit order -- a key per unit, the LSD sort of fga_sort.hip on the index bits, a scan of the hit counts, a gather.
// (Round 3 downloaded units and hits as the kernels left them and ordered them on the host team: 19 + 20 ms per part
// for the 1.9 M units of a 3 Gbp part, 6 + 7 ms for the 1.0 M of the 150 Mbp self comparison.)
// ---------------------------------------------------------------------------------------------------
__global__ void unit_key_kernel(const int64_t *head, int64_t nu, uint4 *rec)
{ const int64_t i = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= nu) return;
  const uint64_t h = (uint64_t) head[i];
  rec[i] = make_uint4((uint32_t) i,(uint32_t) h,(uint32_t) (h >> 32),0u);          // sorted on bits 32.. : the head index
}

__global__ void unit_count_kernel(const uint4 *rec, const fga_unit *units, int64_t nu, int32_t *cnt)
{ const int64_t j = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (j < nu)
    cnt[j] = units[rec[j].x].nhits;
}

// exclusive prefix of n 32-bit counts into 64-bit offsets, three launches: sums of 4096-count tiles, the scan of the sums
// by one workgroup (coalesced, a tile of 1024 sums at a time), the tiles again with their bases.  (One workgroup over
// everything, a contiguous piece per thread, took 2.7 ms for 10^6 counts: every lane of a load on a cache line of its own.)
#define USCAN_TILE 4096
__global__ void __launch_bounds__(256) unit_tile_sum_kernel(const int32_t *cnt, int64_t n, int64_t *tsum)
{ __shared__ int64_t part[4];
  const int64_t base = (int64_t) blockIdx.x*USCAN_TILE;
  int64_t s = 0;
  for (int k = threadIdx.x; k < USCAN_TILE; k += 256)
    if (base+k < n) s += cnt[base+k];
  for (int o = 32; o > 0; o >>= 1)
    s += __shfl_xor(s,o,64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    tsum[blockIdx.x] = part[0]+part[1]+part[2]+part[3];
}

__global__ void __launch_bounds__(1024) unit_sum_scan_kernel(int64_t *tsum, int64_t nt)        // in place, exclusive
{ __shared__ int64_t buf[1024];
  __shared__ int64_t carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int64_t b = 0; b < nt; b += 1024)
    { const int64_t v = b+t < nt ? tsum[b+t] : 0;
      buf[t] = v;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1)
        { const int64_t u = t >= o ? buf[t-o] : 0;
          __syncthreads();
          buf[t] += u;
          __syncthreads();
        }
      if (b+t < nt) tsum[b+t] = carry + buf[t] - v;
      __syncthreads();
      if (t == 1023) carry += buf[1023];
      __syncthreads();
    }
}

__global__ void __launch_bounds__(256) unit_tile_scan_kernel(const int32_t *cnt, int64_t n, const int64_t *tsum, int64_t *out)
{ __shared__ int64_t wsum[4];
  const int64_t base = (int64_t) blockIdx.x*USCAN_TILE + (int64_t) threadIdx.x*16;      // sixteen consecutive counts per thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t c[16];
  int64_t s = 0;
  for (int k = 0; k < 16; k++)
    { c[k] = base+k < n ? cnt[base+k] : 0;
      s += c[k];
    }
  int64_t incl = s;
  for (int o = 1; o < 64; o <<= 1)
    { const int64_t u = __shfl_up(incl,o,64);
      if (lane >= o) incl += u;
    }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int64_t at = tsum[blockIdx.x] + (incl - s);
  for (int w = 0; w < wave; w++)
    at += wsum[w];
  for (int k = 0; k < 16; k++)
    { if (base+k < n) out[base+k] = at;
      at += c[k];
    }
}

void fga_scan_counts(fga_dev *dev, const int32_t *cnt, int64_t n, int64_t *tsum, int64_t *out)
{ const int64_t ntile = (n + USCAN_TILE - 1) / USCAN_TILE;
  if (n <= 0) return;
  hipLaunchKernelGGL(unit_tile_sum_kernel,dim3((unsigned) ntile),dim3(256),0,dev->stream,cnt,n,tsum);
  hipLaunchKernelGGL(unit_sum_scan_kernel,dim3(1),dim3(1024),0,dev->stream,tsum,ntile);
  hipLaunchKernelGGL(unit_tile_scan_kernel,dim3((unsigned) ntile),dim3(256),0,dev->stream,cnt,n,(const int64_t *) tsum,out);
}

__global__ void unit_relay_kernel(const uint4 *rec, const fga_unit *units, const fga_hit *hits, const int64_t *pos,
                                  int64_t nu, fga_unit *ounits, fga_hit *ohits)
{ const int64_t j = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= nu) return;
  fga_unit u = units[rec[j].x];
  const fga_hit *src = hits + u.first_hit;
  fga_hit *dst = ohits + pos[j];
  for (int q = 0; q < u.nhits; q++)
    dst[q] = src[q];
  u.first_hit = pos[j];
  ounits[j] = u;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
extern "C" int fga_chain_scan_device(fga_dev *dev, const fga_dkeys *K, const fga_chain_params *prm, fga_hits **out)
{ *out = NULL;
  FGA_HIP(fga_dev_enter(dev));
  const int64_t n = K->count;
  fga_hits *R = NULL;
  if (n == 0)
    return fga_hits_create(NULL,0,NULL,0,out);

  int nctg = 1 << K->wa;
  int small_limit = SMALL_LIMIT;
  { const char *e = getenv("FGA_CHAIN_SMALL_LIMIT");      // test hook: force units through the wave-parallel kernel
    if (e != NULL && atoi(e) >= 0 && atoi(e) <= SMALL_LIMIT)
      small_limit = atoi(e);
  }
  int64_t hit_cap  = n/8 + 65536;
  int64_t unit_cap = n/8 + 65536;
  int64_t stage_cap = n/8 + 65536;
  const int64_t big_cap = n/(small_limit > 0 ? small_limit : 1) + 16;
  int status = 1;
  void *hbuf = NULL, *ubuf = NULL, *bbuf = NULL, *sbuf = NULL;
  const int stage_slot = (K->slot == SLOT_SORT0) ? SLOT_SORT1 : SLOT_SORT0;     // the sort's idle ping-pong buffer
  int64_t *dalen = NULL;
  unsigned long long hc[CTR_WORDS];

  for (int attempt = 0; attempt < 2; attempt++)
    { const size_t ubytes = sizeof(fga_unit)*(size_t) unit_cap + sizeof(int64_t)*(size_t) unit_cap
                          + sizeof(int64_t)*(size_t) big_cap + sizeof(int64_t)*(size_t) nctg + sizeof(unsigned long long)*CTR_WORDS + 128;
      hbuf = fga_dev_acquire(dev,SLOT_ALNS,sizeof(fga_hit)*(size_t) hit_cap);     // (SLOT_HIST: the unit sort's histograms)
      ubuf = fga_dev_acquire(dev,SLOT_TILES,ubytes);
      if (hbuf == NULL || ubuf == NULL)
        { fga_set_error("fga_chain_scan_device: out of device memory");
          goto fail;
        }
      chain_args A;
      A.keys = K->keys; A.n = n;
      A.wa = K->wa; A.wb = K->wb; A.wd = K->wd; A.wt = K->wt;
      A.s_buck = 12 + K->wt; A.s_b = A.s_buck + K->wd; A.s_a = A.s_b + K->wb; A.s_strand = A.s_a + K->wa;
      A.cbreak = prm->chain_break; A.cmin = prm->chain_min; A.amxpos = prm->amxpos; A.bmxpos = prm->bmxpos;
      A.hits = (fga_hit *) hbuf; A.hit_cap = hit_cap;
      A.units = (fga_unit *) ubuf; A.unit_cap = unit_cap;
      A.unit_head = (int64_t *) (A.units + unit_cap);
      A.bigq = A.unit_head + unit_cap; A.big_cap = big_cap;
      dalen = A.bigq + big_cap;
      A.alen = dalen;
      A.ctr = (unsigned long long *) (dalen + nctg);
      A.small_limit = small_limit;
      // contig lengths by sorted A contig index; indices the table does not cover never occur in a key
      { std::vector<int64_t> al((size_t) nctg,0);
        for (int64_t i = 0; i < nctg && i < prm->nalen; i++) al[(size_t) i] = prm->alen[i];
        if (hipMemcpyAsync(dalen,al.data(),sizeof(int64_t)*(size_t) nctg,hipMemcpyHostToDevice,dev->stream) != hipSuccess ||
            hipMemsetAsync(A.ctr,0,sizeof(unsigned long long)*CTR_WORDS,dev->stream) != hipSuccess ||
            hipStreamSynchronize(dev->stream) != hipSuccess)
          { fga_set_error("fga_chain_scan_device: upload failed");
            goto fail;
          }
      }
      hipEventRecord(dev->ev0,dev->stream);
      { int blk = dev->chain_density > 0. && dev->chain_density < 5e-4 ? 0 : 512;
        const char *ev = getenv("FGA_CHAIN_BLOCK");          // experiments: 256 / 512 / 1024 records per workgroup, 0: sparse kernel
        if (ev != NULL) blk = atoi(ev);
        const int64_t nblk = (n + blk - 1) / (blk > 0 ? blk : 1);
        if (blk == 0)        hipLaunchKernelGGL(chain_sparse_kernel,dim3((unsigned) ((n + 255) / 256)),dim3(256),0,dev->stream,A);
        else if (blk == 256) hipLaunchKernelGGL(chain_small_kernel<256>,dim3((unsigned) nblk),dim3(256),0,dev->stream,A);
        else if (blk == 1024) hipLaunchKernelGGL(chain_small_kernel<1024>,dim3((unsigned) nblk),dim3(1024),0,dev->stream,A);
        else                 hipLaunchKernelGGL(chain_small_kernel<512>,dim3((unsigned) ((n + 511) / 512)),dim3(512),0,dev->stream,A);
      }
      if (hipMemcpyAsync(hc,A.ctr,sizeof(hc),hipMemcpyDeviceToHost,dev->stream) != hipSuccess ||
          hipStreamSynchronize(dev->stream) != hipSuccess)
        { fga_set_error("fga_chain_scan_device: small-unit kernel failed: %s",hipGetErrorString(hipGetLastError()));
          goto fail;
        }
      dev->chain_density = (double) hc[CTR_UNITS] / (double) n;
      if ((int64_t) hc[2] > big_cap)
        { fga_set_error("fga_chain_scan_device: internal error, long-unit queue overflow");
          goto fail;
        }
      if (hc[2] > 0)
        { const int64_t nbig = (int64_t) hc[2];
          big_args B;
          B.seg_cap = nbig + 2*n/SEG_RECS + 16;
          const size_t bbytes = sizeof(big_unit)*(size_t) nbig + sizeof(seg_out)*(size_t) B.seg_cap
                              + sizeof(int32_t)*(size_t) B.seg_cap + 64;
          bbuf = fga_dev_acquire(dev,SLOT_MISC,bbytes);
          sbuf = fga_dev_acquire(dev,stage_slot,sizeof(fga_hit)*(size_t) stage_cap);
          if (bbuf == NULL || sbuf == NULL)
            { fga_set_error("fga_chain_scan_device: out of device memory");
              goto fail;
            }
          B.segs = (seg_out *) bbuf;
          B.bunits = (big_unit *) (B.segs + B.seg_cap);
          B.segmap = (int32_t *) (B.bunits + nbig);
          B.stage = (fga_hit *) sbuf; B.stage_cap = stage_cap;
          hipLaunchKernelGGL(chain_plan_kernel,dim3((unsigned) ((nbig + 63)/64)),dim3(64),0,dev->stream,A,B,nbig);
          if (hipMemcpyAsync(hc,A.ctr,sizeof(hc),hipMemcpyDeviceToHost,dev->stream) != hipSuccess ||
              hipStreamSynchronize(dev->stream) != hipSuccess)
            { fga_set_error("fga_chain_scan_device: plan kernel failed: %s",hipGetErrorString(hipGetLastError()));
              goto fail;
            }
          const int64_t nsegs = (int64_t) hc[4];
          if (nsegs > B.seg_cap)
            { fga_set_error("fga_chain_scan_device: internal error, segment table overflow");
              goto fail;
            }
          if (nsegs > 0)
            { int per_cu = 16;                                     // bench pair: 8 / 16 / 24 / 32 per CU -> 2.0 / 1.6 / 1.7 / 1.6 ms of chain kernels
              { const char *ev = getenv("FGA_CHAIN_SEG_WAVES");     // experiments: persistent wavefronts per CU
                if (ev != NULL && atoi(ev) > 0 && atoi(ev) <= 32) per_cu = atoi(ev);
              }
              int nwg = dev->ncu * per_cu;
              if ((int64_t) nwg > nsegs) nwg = (int) nsegs;
              hipLaunchKernelGGL(chain_segment_kernel,dim3(nwg),dim3(64),0,dev->stream,A,B,nsegs);
              hipLaunchKernelGGL(chain_stitch_kernel,dim3((unsigned) ((nbig + 63)/64)),dim3(64),0,dev->stream,A,B,nbig);
            }
        }
      hipEventRecord(dev->ev1,dev->stream);
      if (hipMemcpyAsync(hc,A.ctr,sizeof(hc),hipMemcpyDeviceToHost,dev->stream) != hipSuccess ||
          hipStreamSynchronize(dev->stream) != hipSuccess)
        { fga_set_error("fga_chain_scan_device: long-unit kernel failed: %s",hipGetErrorString(hipGetLastError()));
          goto fail;
        }
      hipEventElapsedTime(&dev->last_ms[FGA_STAGE_CHAIN],dev->ev0,dev->ev1);
      if ((int64_t) hc[0] <= hit_cap && (int64_t) hc[CTR_UNITS] <= unit_cap && (int64_t) hc[5] <= stage_cap)
        { const int64_t nh = (int64_t) hc[0], nu = (int64_t) hc[CTR_UNITS];
          const double tq0 = fga_wall();
          double tq1;
          uint4 *r0 = NULL, *r1 = NULL, *rs = NULL;
          int32_t *dcnt = NULL; int64_t *dpos = NULL, *dtsum = NULL;
          const int64_t ntile = (nu + USCAN_TILE - 1) / USCAN_TILE;
          fga_unit *ou = NULL; fga_hit *oh = NULL;
          bool bad = false;
          R = (fga_hits *) calloc(1,sizeof(fga_hits));
          if (R != NULL)
            { R->nhits = nh; R->nunits = nu;
              R->hits  = (fga_hit *)  malloc(sizeof(fga_hit)*(size_t) (nh+1));
              R->units = (fga_unit *) malloc(sizeof(fga_unit)*(size_t) (nu+1));
            }
          if (R == NULL || R->hits == NULL || R->units == NULL)
            { fga_hits_free(R); R = NULL; fga_set_error("out of memory"); goto fail; }
          if (nu > 0)
            { int bits = 1;
              while (bits < 63 && ((int64_t) 1 << bits) <= n) bits += 1;      // first-record indices are distinct, in [0,n)
              const unsigned gb = (unsigned) ((nu + 255) / 256);
              if (fga_dmalloc(&r0,sizeof(uint4)*(size_t) nu) != hipSuccess || fga_dmalloc(&r1,sizeof(uint4)*(size_t) nu) != hipSuccess ||
                  fga_dmalloc(&dcnt,sizeof(int32_t)*(size_t) nu) != hipSuccess || fga_dmalloc(&dpos,sizeof(int64_t)*(size_t) nu) != hipSuccess ||
                  fga_dmalloc(&dtsum,sizeof(int64_t)*(size_t) (ntile+1)) != hipSuccess ||
                  fga_dmalloc(&ou,sizeof(fga_unit)*(size_t) nu) != hipSuccess || fga_dmalloc(&oh,sizeof(fga_hit)*(size_t) (nh+1)) != hipSuccess)
                bad = true;
              if (!bad)
                { hipLaunchKernelGGL(unit_key_kernel,dim3(gb),dim3(256),0,dev->stream,(const int64_t *) A.unit_head,nu,r0);
                  if (fga_radix_sort_u128(dev,r0,r1,nu,32,bits,&rs))
                    bad = true;
                }
              if (!bad)
                { hipLaunchKernelGGL(unit_count_kernel,dim3(gb),dim3(256),0,dev->stream,(const uint4 *) rs,(const fga_unit *) A.units,nu,dcnt);
                  fga_scan_counts(dev,dcnt,nu,dtsum,dpos);
                  hipLaunchKernelGGL(unit_relay_kernel,dim3(gb),dim3(256),0,dev->stream,(const uint4 *) rs,(const fga_unit *) A.units,
                                     (const fga_hit *) A.hits,(const int64_t *) dpos,nu,ou,oh);
                  if ((nh > 0 && hipMemcpyAsync(R->hits,oh,sizeof(fga_hit)*(size_t) nh,hipMemcpyDeviceToHost,dev->stream) != hipSuccess) ||
                      hipMemcpyAsync(R->units,ou,sizeof(fga_unit)*(size_t) nu,hipMemcpyDeviceToHost,dev->stream) != hipSuccess ||
                      hipStreamSynchronize(dev->stream) != hipSuccess || hipGetLastError() != hipSuccess)
                    bad = true;
                }
              fga_pool_free(r0); fga_pool_free(r1); fga_pool_free(dcnt); fga_pool_free(dpos); fga_pool_free(dtsum); fga_pool_free(ou); fga_pool_free(oh);
              if (bad)
                { fga_hits_free(R); R = NULL;
                  fga_set_error("fga_chain_scan_device: ordering the units on the device failed");
                  goto fail;
                }
            }
          tq1 = fga_wall();
          if (getenv("FGA_HOST_TIMING") != NULL)
            fprintf(stderr,"chain timing: kernels %.1f ms; %lld hits / %lld units ordered on the device and downloaded in %.1f ms\n",
                    dev->last_ms[FGA_STAGE_CHAIN],(long long) nh,(long long) nu,1e3*(tq1-tq0));
          status = 0;
          break;
        }
      // the outputs did not fit: the counters hold the exact sizes, go again
      // (a stage overflow hides hits from the final count, so every capacity gets the generous bound)
      if ((int64_t) hc[5] > stage_cap)
        hit_cap = unit_cap = stage_cap = n + 16;
      else
        { hit_cap = (int64_t) hc[0] + 16; unit_cap = (int64_t) hc[CTR_UNITS] + 16; }
      fga_dev_release(dev,SLOT_ALNS,hbuf); fga_dev_release(dev,SLOT_TILES,ubuf);
      fga_dev_release(dev,SLOT_MISC,bbuf); fga_dev_release(dev,stage_slot,sbuf);
      hbuf = ubuf = bbuf = sbuf = NULL;
    }
  if (status != 0 && R == NULL)
    fga_set_error("fga_chain_scan_device: output capacity could not be settled");

fail:
  fga_dev_release(dev,SLOT_ALNS,hbuf); fga_dev_release(dev,SLOT_TILES,ubuf);
  fga_dev_release(dev,SLOT_MISC,bbuf); fga_dev_release(dev,stage_slot,sbuf);
  if (status == 0)
    *out = R;
  return status;
}
