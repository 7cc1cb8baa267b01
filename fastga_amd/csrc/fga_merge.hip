// fga_merge.hip -- adaptive k-mer seed merge over two GIX tables on MI355X (gfx950).
//
// Replaces new_merge_thread / new_self_merge_thread / adaptamer_merge of the reference
// (FastGA.c:610-1025, 1616-1909, 2281-2493).  The reference walks both sorted tables with a sequential
// state machine (rcur/rend/eorun/vlcp[]); here the same result is computed as a pure function of
// (T1 entry, T2 panel) -- SURVEY.md Appendix B.1 -- which is what makes it data parallel:
//
//   plen = min(40, max_j LCP(s,c_j)),  R = { j : LCP(s,c_j) >= plen },  |R| >= FREQ -> drop,
//   mask tests, strand rules, one 16-byte seed per surviving (s,c).
//
// Work decomposition ("merge path" over the 2^24-entry prefix index instead of over the entries):
//   cost(p) = idx1[p] + idx2[p] + 2(p+1) is monotone in the 12-mer prefix p; tile w owns the prefixes whose
//   cost falls in (w*TILE, (w+1)*TILE], found by one binary search per tile (merge_partition_kernel).
//   A tile therefore holds <= TILE table entries of T1 and T2 together plus <= TILE/2 prefixes -- unless a
//   single panel is larger than that, in which case the tile is exactly that panel and takes the
//   global-memory path.
// LDS-staged tile (the common case).  Persistent 256-thread workgroups loop over tiles:
//   1. both index slices are loaded as 32-bit offsets relative to the tile start,
//   2. the raw on-disk bytes of both entry ranges stream HBM -> LDS with 16-byte-per-lane coalesced loads
//      (the tables are consumed in their on-disk 13..16-byte width, nothing is re-packed in HBM),
//   3. the panel of every T1 entry comes from a head-flag scatter + block max-scan (no per-entry search);
//      T2 entries are decoded once into 64-bit keys (suffix56 << 8 | mask = byte-swapped first 8 bytes),
//   4. match phase: each lane takes T1 entries (aligned dword LDS reads + v_alignbyte to unpack the odd-width
//      records), lower bound inside the T2 panel, LCP with both neighbours by clz of the key xor, range
//      growth bounded by FREQ; results stay in registers,
//   5. emit phase: one block-wide scan per tile gives every lane its slot in an LDS seed stage, which is
//      flushed to HBM with ONE global atomic per ~1000 seeds and fully coalesced 16-byte stores (a single
//      counter word sustains only ~88 atomics/us on MI355X -- per-wave appends would bound the kernel).
// No MFMA anywhere: integer compare / byte work, HBM-bound by design.

#include "fga_device.hpp"

#ifndef NT
#define NT              256                  // threads per workgroup
#endif
#define NWAVE           (NT/64)
#ifndef TILE_COST
#define TILE_COST       1024                 // cost units per tile
#endif
#define EPT             (TILE_COST/NT)       // T1 entries per thread (upper bound)
#define PCAP            (TILE_COST/2 + 2)    // max prefixes of an LDS tile
#define RAWCAP          (TILE_COST*16 + 96)  // bytes of raw entries staged per tile (E <= 16)
#ifndef STAGE_CAP
#define STAGE_CAP       512                  // seeds staged in LDS between flushes
#endif

static_assert(TILE_COST <= 4*NT,"the owner max-scan handles 4 entries per thread");

#ifdef MERGE_PROF      // per-phase cycle accounting of workgroup thread 0 (tools/merge_bench.py prints it)
__device__ unsigned long long merge_prof[8];
#define PROF_DECL  unsigned long long _pt = clock64(), _pa[8] = {0,0,0,0,0,0,0,0};
#define PROF(k)    { unsigned long long _n = clock64(); _pa[k] += _n - _pt; _pt = _n; }
#define PROF_END   if (threadIdx.x == 0) for (int _k = 0; _k < 8; _k++) atomicAdd(merge_prof+_k,_pa[_k]);
#else
#define PROF_DECL
#define PROF(k)
#define PROF_END
#endif

enum { MODE_PAIR = 0, MODE_FLIP = 1, MODE_SELF = 2 };

struct merge_tile            // 32 bytes
  { int32_t p;               // first prefix of the tile
    int32_t pad;
    int64_t a;               // idx1[p-1]  (entries of T1 before the tile)
    int64_t b;               // idx2[p-1]
    int64_t pad2;
  };

struct merge_args
  { const uint8_t *tab1; const int64_t *idx1;
    const uint8_t *tab2; const int64_t *idx2;
    int   E1, post1, cont1;
    int   E2, post2, cont2;
    int   freq, soft_mask, flip, self;
    int   pbeg, pend;                 // prefix range handled by this call
    int64_t base;                     // cost(pbeg-1)
    int   ntiles;
    int   tile_cost;                  // cost units per tile of this launch's partition
    int   wrawcap;                    // bytes of the wave kernel's raw-entry LDS buffer
    int   pairs;                      // tiles[] holds (begin,end) descriptor pairs of queued tiles; count in *npairs
    const unsigned long long *npairs; int pair_cap;
    const merge_tile *tiles;
    fga_seed *out; int64_t cap;
    unsigned long long *count;        // seeds produced
    unsigned long long *tseed;        // sum of plen (the reference's "ave. len" statistic)
    uint16_t *valid;                  // v3: seeds per 1024-slot block of `out` (holes are left open, consumers skip them)
    int64_t   nblocks;
    unsigned long long *hslots;       // v3: slots left unused
  };

__device__ __forceinline__ int64_t idx_at(const int64_t *idx, int p)     // inclusive cumulative, idx[-1] = 0
{ return p < 0 ? 0 : idx[p]; }

// ---------------------------------------------------------------------------------------------------
// tile boundaries
// ---------------------------------------------------------------------------------------------------
__global__ void merge_partition_kernel(merge_args A, merge_tile *tiles)
{ int w = blockIdx.x*blockDim.x + threadIdx.x;
  if (w > A.ntiles)
    return;
  int p;
  if (w == 0)
    p = A.pbeg;
  else if (w == A.ntiles)
    p = A.pend;
  else
    { int64_t target = A.base + (int64_t) w * A.tile_cost;
      int lo = A.pbeg, hi = A.pend;
      while (lo < hi)
        { int mid = lo + ((hi-lo) >> 1);
          int64_t c = A.idx1[mid] + A.idx2[mid] + 2*((int64_t) mid+1);
          if (c > target) hi = mid; else lo = mid+1;
        }
      p = lo;
    }
  merge_tile t;
  t.p = p; t.pad = 0; t.pad2 = 0;
  t.a = idx_at(A.idx1,p-1);
  t.b = idx_at(A.idx2,p-1);
  tiles[w] = t;
}

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------

// key = suffix56 << 8 | mask.  LCP in bases (>= 12: same panel) of two keys, ignoring the mask byte.
__device__ __forceinline__ int lcp_key(uint64_t a, uint64_t b)
{ uint64_t x = (a ^ b) & ~0xffull;
  return x == 0 ? 40 : 12 + (__clzll((long long) x) >> 1);
}

__device__ __forceinline__ uint32_t bswap32(uint32_t x)
{ return __builtin_bswap32(x); }

// 16 bytes starting at byte offset o of an LDS byte array viewed as aligned dwords
__device__ __forceinline__ void lds_read16(const uint32_t *rawd, uint32_t o, uint32_t &e0, uint32_t &e1,
                                           uint32_t &e2, uint32_t &e3)
{ uint32_t w = o >> 2, sh = o & 3;
  uint32_t d0 = rawd[w], d1 = rawd[w+1], d2 = rawd[w+2], d3 = rawd[w+3], d4 = rawd[w+4];
  e0 = __builtin_amdgcn_alignbyte(d1,d0,sh);
  e1 = __builtin_amdgcn_alignbyte(d2,d1,sh);
  e2 = __builtin_amdgcn_alignbyte(d3,d2,sh);
  e3 = __builtin_amdgcn_alignbyte(d4,d3,sh);
}

__device__ __forceinline__ uint64_t lds_read_key(const uint32_t *rawd, uint32_t o)    // first 8 bytes -> key
{ uint32_t w = o >> 2, sh = o & 3;
  uint32_t d0 = rawd[w], d1 = rawd[w+1], d2 = rawd[w+2];
  uint32_t e0 = __builtin_amdgcn_alignbyte(d1,d0,sh);
  uint32_t e1 = __builtin_amdgcn_alignbyte(d2,d1,sh);
  return ((uint64_t) bswap32(e0) << 32) | bswap32(e1);
}

// payload (bytes 9.. of an entry, little endian) given the entry's dwords e2,e3
__device__ __forceinline__ void split_payload(uint32_t e2, uint32_t e3, int post, int cont,
                                              uint32_t &pos, uint32_t &ctg, uint32_t &sign)
{ uint64_t pv = (((uint64_t) e3 << 32) | e2) >> 8;
  uint32_t pm = post >= 4 ? 0xffffffffu : ((1u << (8*post)) - 1);
  pos = (uint32_t) pv & pm;
  uint32_t c  = (uint32_t) (pv >> (8*post)) & ((1u << (8*cont)) - 1);
  uint32_t sb = 0x80u << (8*(cont-1));
  sign = (c & sb) != 0;
  ctg  = c & (sb-1);
}

// pull the 64-byte line at p towards L2 (the tile load proper follows an iteration later).  The byte is a real load
// whose value is folded into a dummy accumulator only AFTER the next tile's own loads have been issued, so the
// wait for it coincides with the wait the tile load needs anyway.
__device__ __forceinline__ uint32_t l2_touch(const uint8_t *p)
{ return *(const volatile uint8_t *) p; }

__device__ __forceinline__ int wave_incl_scan_add(int v)
{ int lane = threadIdx.x & 63;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { int y = __shfl_up(v,d,64);
      if (lane >= d) v += y;
    }
  return v;
}

__device__ __forceinline__ int wave_incl_scan_max(int v)
{ int lane = threadIdx.x & 63;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { int y = __shfl_up(v,d,64);
      if (lane >= d) v = v > y ? v : y;
    }
  return v;
}

struct stage_t
  { fga_seed *buf;               // LDS, STAGE_CAP seeds
    int      *n;                 // LDS, seeds currently staged
    int      *wtot;              // LDS, NWAVE per-wave totals
    unsigned long long *gbase;   // LDS, base returned by the flush atomic
  };

// flush the LDS stage to HBM: one atomic, coalesced 16-byte stores.  All threads must call it.
__device__ __forceinline__ void stage_flush(const merge_args &A, const stage_t &S)
{ int n = *S.n;
  if (n == 0)
    return;
  if (threadIdx.x == 0)
    *S.gbase = atomicAdd(A.count,(unsigned long long) n);
  __syncthreads();
  int64_t base = (int64_t) *S.gbase;
  for (int x = threadIdx.x; x < n; x += NT)
    if (base + x < A.cap)
      A.out[base + x] = S.buf[x];
  __syncthreads();
  if (threadIdx.x == 0)
    *S.n = 0;
  __syncthreads();
}

// Block-wide slot assignment for `cnt` seeds per thread.  Returns where this thread writes:
//   dst = LDS stage (direct == false) or HBM (direct == true), at index `at`.
__device__ __forceinline__ void block_slots(const merge_args &A, const stage_t &S, int cnt,
                                            bool &any, bool &direct, int64_t &at)
{ const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int inc = wave_incl_scan_add(cnt);
  if (lane == 63)
    S.wtot[wave] = inc;
  __syncthreads();
  int btotal = 0, wbase = 0;
  #pragma unroll
  for (int w = 0; w < NWAVE; w++)
    { int t = S.wtot[w];
      if (w < wave) wbase += t;
      btotal += t;
    }
  any = btotal > 0;
  direct = false;
  at = 0;
  if (!any)
    { __syncthreads();
      return;
    }
  if (*S.n + btotal > STAGE_CAP)
    stage_flush(A,S);
  if (btotal > STAGE_CAP)
    { if (threadIdx.x == 0)
        *S.gbase = atomicAdd(A.count,(unsigned long long) btotal);
      __syncthreads();
      direct = true;
      at = (int64_t) *S.gbase + wbase + (inc - cnt);
      __syncthreads();
      return;
    }
  at = *S.n + wbase + (inc - cnt);
  __syncthreads();
  if (threadIdx.x == 0)
    *S.n += btotal;
  // the caller synchronises before the stage is read or flushed again
}

template <int MODE>
__device__ __forceinline__ fga_seed make_seed(int plen, uint32_t spos, uint32_t sctg, uint32_t ssign,
                                              uint32_t cpos, uint32_t cctg, uint32_t csign)
{ fga_seed sd;
  if (MODE == MODE_FLIP)          // table 1 is genome 2: A side = c (forward), B side = s
    { sd.apos = cpos; sd.bpos = spos;
      sd.actg = (cctg << 8) | (uint32_t) plen;
      sd.bctg = sctg | (ssign << 30) | (ssign << 31);
    }
  else if (MODE == MODE_SELF)     // stream N iff the signs agree; A payload goes out with its sign cleared
    { sd.apos = spos; sd.bpos = cpos;
      sd.actg = (sctg << 8) | (uint32_t) plen;
      sd.bctg = cctg | (csign << 30) | ((uint32_t) (ssign != csign) << 31);
    }
  else
    { sd.apos = spos; sd.bpos = cpos;
      sd.actg = (sctg << 8) | (uint32_t) plen;
      sd.bctg = cctg | (csign << 30) | (csign << 31);
    }
  return sd;
}

// ---------------------------------------------------------------------------------------------------
// Oversize tiles: same semantics straight from HBM (byte loads; rare)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t glb_key(const uint8_t *e)
{ return ((uint64_t) e[0] << 56) | ((uint64_t) e[1] << 48) | ((uint64_t) e[2] << 40) | ((uint64_t) e[3] << 32)
       | ((uint64_t) e[4] << 24) | ((uint64_t) e[5] << 16) | ((uint64_t) e[6] << 8) | (uint64_t) e[7];
}

__device__ __forceinline__ void glb_payload(const uint8_t *e, int post, int cont,
                                            uint32_t &pos, uint32_t &ctg, uint32_t &sign)
{ uint32_t p = 0, c = 0;
  for (int k = 0; k < post; k++) p |= (uint32_t) e[9+k] << (8*k);
  for (int k = 0; k < cont; k++) c |= (uint32_t) e[9+post+k] << (8*k);
  uint32_t sb = 0x80u << (8*(cont-1));
  pos = p; sign = (c & sb) != 0; ctg = c & (sb-1);
}

template <int MODE>
__device__ void global_tile(const merge_args &A, const stage_t &S, int p0, int p1, int64_t a0, int64_t a1,
                            unsigned long long &tsum)
{ const uint8_t *tab2 = (MODE == MODE_SELF) ? A.tab1 : A.tab2;
  const int64_t *idx2 = (MODE == MODE_SELF) ? A.idx1 : A.idx2;
  const int E1 = A.E1, E2 = A.E2;
  const int64_t rounds = (a1 - a0 + NT - 1) / NT;
  for (int64_t r = 0; r < rounds; r++)
    { int64_t i = a0 + r*NT + threadIdx.x;
      int cnt = 0, plen = 0;
      int64_t low = 0, hgh = 0;
      uint32_t spos = 0, sctg = 0, ssign = 0;
      int mlen = 41;
      if (i < a1)
        { int lo = p0, hi = p1-1;                   // smallest p with idx1[p] > i
          while (lo < hi)
            { int m = lo + ((hi-lo) >> 1);
              if (A.idx1[m] > i) hi = m; else lo = m+1;
            }
          int64_t b0 = idx_at(idx2,lo-1), b1 = idx2[lo];
          const uint8_t *e = A.tab1 + i*E1;
          glb_payload(e,A.post1,A.cont1,spos,sctg,ssign);
          bool go = (b1 > b0) && !(MODE == MODE_PAIR && ssign);
          if (go)
            { uint64_t ks = glb_key(e);
              int64_t lb;
              if (MODE == MODE_SELF)
                { int lk  = (i > b0)   ? lcp_key(ks,glb_key(tab2 + (i-1)*E2)) : 0;
                  int lk1 = (i+1 < b1) ? lcp_key(ks,glb_key(tab2 + (i+1)*E2)) : 11;
                  plen = lk > lk1 ? lk : lk1;
                  low = i; hgh = i+1; lb = i;
                }
              else
                { int64_t l = b0, h = b1;
                  uint64_t kq = ks & ~0xffull;
                  while (l < h)
                    { int64_t m = (l+h) >> 1;
                      if (glb_key(tab2 + m*E2) < kq) l = m+1; else h = m;
                    }
                  int la = (l > b0) ? lcp_key(ks,glb_key(tab2 + (l-1)*E2)) : 0;
                  int lc = (l < b1) ? lcp_key(ks,glb_key(tab2 + l*E2)) : 0;
                  plen = la > lc ? la : lc;
                  low = hgh = lb = l;
                }
              while (low > b0 && lb-low <= A.freq && lcp_key(ks,glb_key(tab2 + (low-1)*E2)) >= plen)
                low -= 1;
              while (hgh < b1 && hgh-low <= A.freq && lcp_key(ks,glb_key(tab2 + hgh*E2)) >= plen)
                hgh += 1;
              mlen = A.soft_mask ? plen : 41;
              if (hgh-low < A.freq && (int) (ks & 0xff) < mlen)
                for (int64_t j = low; j < hgh; j++)
                  { const uint8_t *c = tab2 + j*E2;
                    if ((MODE == MODE_SELF && j == i) || c[7] >= mlen)
                      continue;
                    if (MODE == MODE_FLIP)
                      { uint32_t cp, cc, cs;
                        glb_payload(c,A.post2,A.cont2,cp,cc,cs);
                        if (cs) continue;
                      }
                    cnt += 1;
                  }
            }
        }
      bool any, direct;
      int64_t at;
      block_slots(A,S,cnt,any,direct,at);
      if (!any)
        continue;
      if (cnt > 0)
        { tsum += (unsigned long long) cnt * plen;
          for (int64_t j = low; j < hgh; j++)
            { const uint8_t *c = tab2 + j*E2;
              if ((MODE == MODE_SELF && j == i) || c[7] >= mlen)
                continue;
              uint32_t cp, cc, cs;
              glb_payload(c,A.post2,A.cont2,cp,cc,cs);
              if (MODE == MODE_FLIP && cs)
                continue;
              fga_seed sd = make_seed<MODE>(plen,spos,sctg,ssign,cp,cc,cs);
              if (direct)
                { if (at < A.cap) A.out[at] = sd; }
              else
                S.buf[at] = sd;
              at += 1;
            }
        }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// the merge kernel
// ---------------------------------------------------------------------------------------------------
#ifdef MERGE_WAVES_PER_EU          // occupancy experiments: cap the VGPR budget so that this many waves fit a SIMD
#define MERGE_OCC __attribute__((amdgpu_waves_per_eu(MERGE_WAVES_PER_EU,MERGE_WAVES_PER_EU)))
#else
#define MERGE_OCC
#endif

template <int MODE>
__global__ __launch_bounds__(NT) MERGE_OCC
void seed_merge_kernel(merge_args A)
{ __shared__ uint32_t la[PCAP+1];            // la[q] = #T1 entries of the tile in prefixes <= p0+q
  __shared__ uint32_t lb[PCAP+1];
  __shared__ __attribute__((aligned(16))) uint8_t  raw[RAWCAP];
  __shared__ __attribute__((aligned(16))) uint64_t keyB[TILE_COST];
  __shared__ __attribute__((aligned(16))) uint16_t own[TILE_COST];
  __shared__ __attribute__((aligned(16))) fga_seed stagebuf[STAGE_CAP];
  __shared__ int stage_n;
  __shared__ int wtot[NWAVE];
  __shared__ unsigned long long gbase;

  const int tid = threadIdx.x;
  stage_t S;
  S.buf = stagebuf; S.n = &stage_n; S.wtot = wtot; S.gbase = &gbase;
  if (tid == 0)
    stage_n = 0;
  unsigned long long tsum = 0;
  __syncthreads();
  PROF_DECL

  const int E1 = A.E1, E2 = A.E2;
  const int freq = A.freq;

  // the descriptors of the NEXT tile are fetched one iteration ahead and its table bytes / index slices are pulled
  // into L2 with throw-away loads, so that the tile load at the top of an iteration is one L2 round trip instead
  // of two dependent HBM round trips
  uint32_t pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0, pfacc = 0;
  const int step = A.pairs ? 2 : 1;
  int ntl = A.ntiles;
  if (A.pairs)
    { const unsigned long long np_ = *A.npairs;
      ntl = 2 * (int) (np_ < (unsigned long long) A.pair_cap ? np_ : (unsigned long long) A.pair_cap);
    }
  merge_tile nt0 = A.tiles[(int) blockIdx.x*step < ntl ? blockIdx.x*step : 0];
  merge_tile nt1 = A.tiles[(int) blockIdx.x*step < ntl ? blockIdx.x*step+1 : 0];
  for (int tile = blockIdx.x*step; tile < ntl; tile += gridDim.x*step)
    { const merge_tile t0 = nt0;
      const merge_tile t1 = nt1;
      const uint32_t ppf = pf0 + pf1 + pf2 + pf3;        // touched for THIS tile an iteration ago
      { const int nx = tile + gridDim.x*step;
        if (nx < ntl)
          { nt0 = A.tiles[nx]; nt1 = A.tiles[nx+1];
            const int64_t q1 = nt0.a*E1, r1 = nt1.a*E1;
            const int64_t q2 = ((MODE == MODE_SELF) ? nt0.a : nt0.b)*E2, r2 = ((MODE == MODE_SELF) ? nt1.a : nt1.b)*E2;
            const int np2 = nt1.p - nt0.p;
            // one byte per 64-byte line and lane; the value is never used
            int64_t off = (q1 & ~(int64_t) 63) + 64*(int64_t) tid;
            pf0 = pf1 = pf2 = pf3 = 0;
            if (off < r1 && r1 - q1 <= RAWCAP)
              pf0 = l2_touch(A.tab1 + off);
            if (MODE != MODE_SELF)
              { off = (q2 & ~(int64_t) 63) + 64*(int64_t) tid;
                if (off < r2 && r2 - q2 <= RAWCAP)
                  pf1 = l2_touch(A.tab2 + off);
              }
            if (tid*8 < np2 && np2 <= PCAP)
              { pf2 = l2_touch((const uint8_t *) (A.idx1 + nt0.p + tid*8));
                if (MODE != MODE_SELF)
                  pf3 = l2_touch((const uint8_t *) (A.idx2 + nt0.p + tid*8));
              }
          }
      }
      const int p0 = t0.p, p1 = t1.p;
      const int np = p1 - p0;
      if (np <= 0)
        continue;
      const int64_t a0 = t0.a, a1 = t1.a;
      const int64_t b0 = (MODE == MODE_SELF) ? a0 : t0.b, b1 = (MODE == MODE_SELF) ? a1 : t1.b;
      const int64_t n1l = a1 - a0, n2l = b1 - b0;
      if (n1l == 0 || n2l == 0)
        continue;

      // byte extents, aligned down to 16 for the coalesced copy
      const int64_t s1 = a0*E1, e1 = a1*E1;
      const int64_t s2 = b0*E2, e2 = b1*E2;
      const int64_t s1a = s1 & ~(int64_t) 15, s2a = s2 & ~(int64_t) 15;
      const int64_t len1 = ((e1 - s1a) + 15) & ~(int64_t) 15;
      const int64_t len2 = (MODE == MODE_SELF) ? 0 : (((e2 - s2a) + 15) & ~(int64_t) 15);

      if (np > PCAP-1 || n1l + n2l > TILE_COST || len1 + len2 > RAWCAP - 32)
        { global_tile<MODE>(A,S,p0,p1,a0,a1,tsum);
          __syncthreads();
          continue;
        }
      const int n1 = (int) n1l, n2 = (int) n2l;

      // 1. raw bytes HBM -> LDS, 16 B per lane; index slices; clear the owner array
      { const uint4 *g1 = (const uint4 *) (A.tab1 + s1a);
        uint4 *l1 = (uint4 *) raw;
        const int n16 = (int) (len1 >> 4);
        for (int x = tid; x < n16; x += NT)           // global_load_lds: a wavefront's 64 lanes write 1 KB in lane order
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g1 + x),
                                           (__attribute__((address_space(3))) void *) (l1 + x),16,0,0);
        if (MODE != MODE_SELF)
          { const uint4 *g2 = (const uint4 *) (A.tab2 + s2a);
            uint4 *l2 = (uint4 *) (raw + len1);
            const int m16 = (int) (len2 >> 4);
            for (int x = tid; x < m16; x += NT)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g2 + x),
                                               (__attribute__((address_space(3))) void *) (l2 + x),16,0,0);
          }
      }
      for (int q = tid; q < np; q += NT)
        { la[q] = (uint32_t) (A.idx1[p0+q] - a0);
          lb[q] = (MODE == MODE_SELF) ? la[q] : (uint32_t) (A.idx2[p0+q] - b0);
        }
      pfacc += ppf;
      { uint2 *o2 = (uint2 *) own;
        for (int x = tid; x < (n1+3)/4; x += NT)
          o2[x] = make_uint2(0,0);
      }
      __syncthreads();
      PROF(0)

      const uint32_t *rawd = (const uint32_t *) raw;
      const uint32_t o1 = (uint32_t) (s1 - s1a);
      const uint32_t o2 = (MODE == MODE_SELF) ? o1 : (uint32_t) (len1 + (s2 - s2a));

      // 2. head flags of the non-empty T1 panels; T2 keys
      for (int q = tid; q < np; q += NT)
        { uint32_t s = q ? la[q-1] : 0;
          if (la[q] > s)
            own[s] = (uint16_t) q;
        }
      for (int j = tid; j < n2; j += NT)
        keyB[j] = lds_read_key(rawd,o2 + (uint32_t) j*E2);
      __syncthreads();
      PROF(1)

      // 3. owner[i] = max head at or before i (block max-scan, 4 consecutive entries per thread)
      int nlive;
      uint32_t *clist = (uint32_t *) (keyB + n2);
      { uint2 v = ((uint2 *) own)[tid];
        int x0 = v.x & 0xffff, x1 = v.x >> 16, x2 = v.y & 0xffff, x3 = v.y >> 16;
        x1 = x1 > x0 ? x1 : x0;
        x2 = x2 > x1 ? x2 : x1;
        x3 = x3 > x2 ? x3 : x2;
        int inc = wave_incl_scan_max(x3);
        if ((tid & 63) == 63)
          wtot[tid >> 6] = inc;
        __syncthreads();
        int prev = __shfl_up(inc,1,64);
        if ((tid & 63) == 0) prev = 0;
        #pragma unroll
        for (int w = 0; w < NWAVE; w++)
          if (w < (tid >> 6))
            { int t = wtot[w];
              prev = prev > t ? prev : t;
            }
        x0 = x0 > prev ? x0 : prev;
        x1 = x1 > prev ? x1 : prev;
        x2 = x2 > prev ? x2 : prev;
        x3 = x3 > prev ? x3 : prev;
        // 3b. compaction: only the T1 entries that can emit -- forward strand in the plain pass (FastGA.c:921-928)
        //     and a non-empty T2 panel -- go on to the match phase, packed (i | q << 16) behind the T2 keys
        //     (8 n2 + 4 n1 <= 8 TILE_COST bytes), so that its rounds run with full wavefronts instead of half-empty ones
        int live = 0;
        uint32_t pk[4];
        { const int xs[4] = { x0, x1, x2, x3 };
          #pragma unroll
          for (int e = 0; e < 4; e++)
            { const int i = tid*4 + e, q = xs[e];
              bool ok = i < n1;
              if (ok)
                { const int pb0 = q ? (int) lb[q-1] : 0, pb1 = (int) lb[q];
                  ok = pb1 > pb0;
                  if (ok && MODE == MODE_PAIR)
                    { const uint32_t sb = o1 + (uint32_t) i*E1 + E1 - 1;
                      ok = !((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80);
                    }
                }
              pk[e] = ok ? ((uint32_t) i | ((uint32_t) q << 16)) : 0xffffffffu;
              live += ok;
            }
        }
        const int linc = wave_incl_scan_add(live);
        __syncthreads();                                  // everybody has read wtot[] (the max scan) by now
        if ((tid & 63) == 63)
          wtot[tid >> 6] = linc;
        __syncthreads();
        int lbase = linc - live;
        nlive = 0;
        #pragma unroll
        for (int w = 0; w < NWAVE; w++)
          { const int t = wtot[w];
            if (w < (tid >> 6)) lbase += t;
            nlive += t;
          }
        #pragma unroll
        for (int e = 0; e < 4; e++)
          if (pk[e] != 0xffffffffu)
            clist[lbase++] = pk[e];
        __syncthreads();
      }

      PROF(2)
      // 4. match phase: T1 entries tid, tid+NT, ...; results packed in registers
      int      r_low[EPT], r_cnt[EPT], r_plen[EPT], r_i[EPT];
      int      total = 0;
      #pragma unroll
      for (int r = 0; r < EPT; r++)
        { const int c = r*NT + tid;
          r_cnt[r] = 0; r_low[r] = 0; r_plen[r] = 0; r_i[r] = 0;
          if (c >= nlive)
            continue;
          const uint32_t ce = clist[c];
          const int i = (int) (ce & 0xffff), q = (int) (ce >> 16);
          r_i[r] = i;
          const uint32_t oe = o1 + (uint32_t) i*E1;
          const int pb0 = q ? (int) lb[q-1] : 0, pb1 = (int) lb[q];
          const uint64_t ks = (MODE == MODE_SELF) ? keyB[i] : lds_read_key(rawd,oe);
          int low, hgh, plen, lbnd;
          if (MODE == MODE_SELF)
            { int lk  = (i > pb0)   ? lcp_key(ks,keyB[i-1]) : 0;
              int lk1 = (i+1 < pb1) ? lcp_key(ks,keyB[i+1]) : 11;
              plen = lk > lk1 ? lk : lk1;
              low = i; hgh = i+1; lbnd = i;
            }
          else
            { int lo = pb0, hi = pb1;
              const uint64_t kq = ks & ~0xffull;
              while (lo < hi)
                { int m = (lo+hi) >> 1;
                  if (keyB[m] < kq) lo = m+1; else hi = m;
                }
              int la_ = (lo > pb0) ? lcp_key(ks,keyB[lo-1]) : 0;
              int lc_ = (lo < pb1) ? lcp_key(ks,keyB[lo]) : 0;
              plen = la_ > lc_ ? la_ : lc_;
              low = hgh = lbnd = lo;
            }
          while (low > pb0 && lbnd-low <= freq && lcp_key(ks,keyB[low-1]) >= plen)
            low -= 1;
          while (hgh < pb1 && hgh-low <= freq && lcp_key(ks,keyB[hgh]) >= plen)
            hgh += 1;
          if (hgh-low >= freq)
            continue;
          const int mlen = A.soft_mask ? plen : 41;
          if ((int) (ks & 0xff) >= mlen)
            continue;
          int cnt;
          if (MODE == MODE_FLIP || A.soft_mask)
            { cnt = 0;
              for (int j = low; j < hgh; j++)
                { if ((int) (keyB[j] & 0xff) >= mlen)
                    continue;
                  if (MODE == MODE_FLIP)
                    { uint32_t sb = o2 + (uint32_t) j*E2 + E2 - 1;
                      if ((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80)
                        continue;
                    }
                  if (MODE == MODE_SELF && j == i)
                    continue;
                  cnt += 1;
                }
            }
          else
            cnt = (hgh-low) - (MODE == MODE_SELF ? 1 : 0);
          r_cnt[r] = cnt; r_low[r] = low | (hgh << 16); r_plen[r] = plen;
          total += cnt;
          tsum  += (unsigned long long) cnt * plen;
        }

      PROF(3)
      // 5. emit phase
      bool any, direct;
      int64_t at;
      block_slots(A,S,total,any,direct,at);
      PROF(4)
      if (any)
        { if (total > 0)
            { const int mfull = A.soft_mask;
              #pragma unroll
              for (int r = 0; r < EPT; r++)
                { if (r_cnt[r] == 0)
                    continue;
                  const int i = r_i[r];
                  const int low = r_low[r] & 0xffff, hgh = r_low[r] >> 16, plen = r_plen[r];
                  const int mlen = mfull ? plen : 41;
                  uint32_t e0, e1_, e2_, e3, spos, sctg, ssign;
                  lds_read16(rawd,o1 + (uint32_t) i*E1,e0,e1_,e2_,e3);
                  split_payload(e2_,e3,A.post1,A.cont1,spos,sctg,ssign);
                  for (int j = low; j < hgh; j++)
                    { if ((int) (keyB[j] & 0xff) >= mlen)
                        continue;
                      if (MODE == MODE_SELF && j == i)
                        continue;
                      uint32_t c0, c1, c2, c3, cpos, cctg, csign;
                      lds_read16(rawd,o2 + (uint32_t) j*E2,c0,c1,c2,c3);
                      split_payload(c2,c3,A.post2,A.cont2,cpos,cctg,csign);
                      if (MODE == MODE_FLIP && csign)
                        continue;
                      fga_seed sd = make_seed<MODE>(plen,spos,sctg,ssign,cpos,cctg,csign);
                      if (direct)
                        { if (at < A.cap) A.out[at] = sd; }
                      else
                        stagebuf[at] = sd;
                      at += 1;
                    }
                }
            }
        }
      __syncthreads();     // tile buffers and the stage are reused by the next tile
      PROF(5)
    }
  PROF_END

  __syncthreads();
  stage_flush(A,S);
  asm volatile("" :: "v"(pfacc));          // keeps the L2 touches alive
  // sum of plen: one atomic per wave at the very end
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if ((tid & 63) == 0 && tsum != 0)
    atomicAdd(A.tseed,tsum);
}

// ---------------------------------------------------------------------------------------------------
// the wave-per-tile merge kernel
// ---------------------------------------------------------------------------------------------------
// Same tile algorithm, but one wavefront owns one (4x smaller) tile and never meets another wavefront: no workgroup
// barrier, no block scan, no shared stage.  Steps are separated by wavefront-scope fences only; scans are DPP / mbcnt.
// Output: a wavefront reserves CHUNK_SEEDS slots of the seed buffer with ONE global atomic and fills them over
// many tiles with direct 16-byte stores; a tile that does not fit the rest of the chunk spills into the next one, so
// the only unused slots are the tail of each wavefront's last chunk.  Those "holes" are recorded and closed afterwards
// by moving the seeds at the end of the buffer into them (seed order is irrelevant: the sort follows).
// Tiles that do not fit the per-wave LDS budget (a single k-mer panel of more than WTILE_COST entries) are queued and
// go through the workgroup kernel above.
#ifndef WTILE_COST
#define WTILE_COST      256
#endif
static_assert(WTILE_COST == 256,"the per-lane owner scan, the owner clear and the emission descriptors assume 4 entries per lane and 8-bit tile indices");
#define WEPT            (WTILE_COST/64)
#define WPCAP           (WTILE_COST/2 + 2)
#define WRAWCAP         (WTILE_COST*16 + 96)
#ifndef CHUNK_SEEDS
#define CHUNK_SEEDS     1024
#endif

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL,"wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define T2_LCP(j) ((rawd[(o2 + (uint32_t) (j)*E2 + 8) >> 2] >> (8*((o2 + (uint32_t) (j)*E2 + 8) & 3))) & 0xff)

struct wave_out
  { unsigned long long *holes;      // [2*hole_cap] begin,end of unused slot ranges
    unsigned long long *ctr;        // [0] holes, [1] queued oversize tiles
    int                *bigq;
    int                 hole_cap, big_cap;
  };

__device__ __forceinline__ int wave_excl_scan_add_dpp(int v, int &total)
{ int x = v, t;
  t = __builtin_amdgcn_update_dpp(0,x,0x111,0xf,0xf,true); x += t;
  t = __builtin_amdgcn_update_dpp(0,x,0x112,0xf,0xf,true); x += t;
  t = __builtin_amdgcn_update_dpp(0,x,0x114,0xf,0xf,true); x += t;
  t = __builtin_amdgcn_update_dpp(0,x,0x118,0xf,0xf,true); x += t;
  t = __builtin_amdgcn_update_dpp(0,x,0x142,0xa,0xf,false); x += t;
  t = __builtin_amdgcn_update_dpp(0,x,0x143,0xc,0xf,false); x += t;
  total = __builtin_amdgcn_readlane(x,63);
  return x - v;
}

__device__ __forceinline__ int wave_incl_scan_max_dpp(int v)      // v >= 0
{ int x = v, t;
  t = __builtin_amdgcn_update_dpp(0,x,0x111,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x112,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x114,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x118,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x142,0xa,0xf,false); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x143,0xc,0xf,false); x = x > t ? x : t;
  return x;
}

#ifndef WAVE_OCC
#define WAVE_OCC 5                      // resident wavefronts per SIMD the register budget is held to
#endif

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVE_OCC,WAVE_OCC)))
void seed_merge_wave_kernel(merge_args A, wave_out W)
{ __shared__ uint16_t la[WPCAP+1];
  __shared__ uint16_t lb[WPCAP+1];
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];      // A.wrawcap bytes (sized by the entry widths)
  __shared__ __attribute__((aligned(16))) uint64_t keyB[WTILE_COST];
  __shared__ __attribute__((aligned(16))) uint16_t own[WTILE_COST];

  const int lane = threadIdx.x;
  const int E1 = A.E1, E2 = A.E2;
  const int freq = A.freq;
  unsigned long long tsum = 0;
  int64_t chunk_pos = 0, chunk_end = 0;         // this wavefront's current output chunk (wave-uniform)

  merge_tile nt0 = A.tiles[blockIdx.x < (unsigned) A.ntiles ? blockIdx.x : 0];
  merge_tile nt1 = A.tiles[blockIdx.x < (unsigned) A.ntiles ? blockIdx.x+1 : 0];
  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x)
    { const merge_tile t0 = nt0;
      const merge_tile t1 = nt1;
      { const int nx = tile + gridDim.x;
        if (nx < A.ntiles)
          { nt0 = A.tiles[nx]; nt1 = A.tiles[nx+1];
          }
      }
      const int p0 = t0.p, p1 = t1.p;
      const int np = p1 - p0;
      if (np <= 0)
        continue;
      const int64_t a0 = t0.a, a1 = t1.a;
      const int64_t b0 = (MODE == MODE_SELF) ? a0 : t0.b, b1 = (MODE == MODE_SELF) ? a1 : t1.b;
      const int64_t n1l = a1 - a0, n2l = b1 - b0;
      if (n1l == 0 || n2l == 0)
        continue;
      const int64_t s1 = a0*E1, e1 = a1*E1;
      const int64_t s2 = b0*E2, e2 = b1*E2;
      const int64_t s1a = s1 & ~(int64_t) 15, s2a = s2 & ~(int64_t) 15;
      const int64_t len1 = ((e1 - s1a) + 15) & ~(int64_t) 15;
      const int64_t len2 = (MODE == MODE_SELF) ? 0 : (((e2 - s2a) + 15) & ~(int64_t) 15);
      if (np > WPCAP-1 || n1l + n2l > WTILE_COST || len1 + len2 > A.wrawcap - 32)
        { if (lane == 0)                            // oversize: the workgroup kernel takes it afterwards
            { const unsigned long long q = atomicAdd(W.ctr+1,1ull);
              if ((int64_t) q < W.big_cap)
                W.bigq[q] = tile;
            }
          continue;
        }
      const int n1 = (int) n1l, n2 = (int) n2l;

      // 1. raw bytes HBM -> LDS, index slices, owner array cleared
      // straight into LDS (global_load_lds_dwordx4: destination = the first lane's address + lane x 16, no VGPR staging)
      { const uint4 *g1 = (const uint4 *) (A.tab1 + s1a);
        uint4 *l1 = (uint4 *) raw;
        const int n16 = (int) (len1 >> 4);
        for (int x = lane; x < n16; x += 64)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g1 + x),
                                           (__attribute__((address_space(3))) void *) (l1 + x),16,0,0);
        if (MODE != MODE_SELF)
          { const uint4 *g2 = (const uint4 *) (A.tab2 + s2a);
            uint4 *l2 = (uint4 *) (raw + len1);
            const int m16 = (int) (len2 >> 4);
            for (int x = lane; x < m16; x += 64)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g2 + x),
                                               (__attribute__((address_space(3))) void *) (l2 + x),16,0,0);
          }
      }
      for (int q = lane; q < np; q += 64)
        { la[q] = (uint16_t) (A.idx1[p0+q] - a0);
          lb[q] = (MODE == MODE_SELF) ? la[q] : (uint16_t) (A.idx2[p0+q] - b0);
        }
      ((uint2 *) own)[lane] = make_uint2(0,0);       // WTILE_COST = 4 entries per lane
      WSYNC();
#if defined(KNOCK_AFTER_LOAD)                        // phase knock-outs: timing experiments only (wrong output)
      continue;
#endif

      const uint32_t *rawd = (const uint32_t *) raw;
      const uint32_t o1 = (uint32_t) (s1 - s1a);
      const uint32_t o2 = (MODE == MODE_SELF) ? o1 : (uint32_t) (len1 + (s2 - s2a));

      // 2. head flags of the non-empty T1 panels; T2 keys
      for (int q = lane; q < np; q += 64)
        { const uint32_t s = q ? la[q-1] : 0;
          if (la[q] > s)
            own[s] = (uint16_t) q;
        }
      for (int j = lane; j < n2; j += 64)
        keyB[j] = lds_read_key(rawd,o2 + (uint32_t) j*E2);
      WSYNC();

      // 3. owner of every T1 entry (wave max-scan, 4 consecutive entries per lane) and compaction of the entries that
      //    can emit, packed (i | q << 16) behind the T2 keys
      int nlive;
      uint32_t *clist = (uint32_t *) (keyB + n2);
      { const uint2 v = ((const uint2 *) own)[lane];
        int x0 = v.x & 0xffff, x1 = v.x >> 16, x2 = v.y & 0xffff, x3 = v.y >> 16;
        x1 = x1 > x0 ? x1 : x0;
        x2 = x2 > x1 ? x2 : x1;
        x3 = x3 > x2 ? x3 : x2;
        const int inc = wave_incl_scan_max_dpp(x3);
        const int prev = __builtin_amdgcn_update_dpp(0,inc,0x138,0xf,0xf,false);      // lane-1's inclusive value, 0 for lane 0
        x0 = x0 > prev ? x0 : prev;
        x1 = x1 > prev ? x1 : prev;
        x2 = x2 > prev ? x2 : prev;
        x3 = x3 > prev ? x3 : prev;
        int live = 0;
        uint32_t pk[4];
        const int xs[4] = { x0, x1, x2, x3 };
        #pragma unroll
        for (int e = 0; e < 4; e++)
          { const int i = lane*4 + e, q = xs[e];
            bool ok = i < n1;
            if (ok)
              { const int pb0 = q ? (int) lb[q-1] : 0, pb1 = (int) lb[q];
                ok = pb1 > pb0;
                if (ok && MODE == MODE_PAIR)
                  { const uint32_t sb = o1 + (uint32_t) i*E1 + E1 - 1;
                    ok = !((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80);
                  }
              }
            pk[e] = ok ? ((uint32_t) i | ((uint32_t) q << 16)) : 0xffffffffu;
            live += ok;
          }
        int lbase = wave_excl_scan_add_dpp(live,nlive);
        #pragma unroll
        for (int e = 0; e < 4; e++)
          if (pk[e] != 0xffffffffu)
            clist[lbase++] = pk[e];
      }
      WSYNC();
#if defined(KNOCK_AFTER_COMPACT)
      continue;
#endif

      // 4. match phase
      int r_low[WEPT], r_cnt[WEPT], r_plen[WEPT], r_i[WEPT];
      int total = 0;
      #pragma unroll
      for (int r = 0; r < WEPT; r++)
        { const int c = r*64 + lane;
          r_cnt[r] = 0; r_low[r] = 0; r_plen[r] = 0; r_i[r] = 0;
          if (c >= nlive)
            continue;
          const uint32_t ce = clist[c];
          const int i = (int) (ce & 0xffff), q = (int) (ce >> 16);
          r_i[r] = i;
          const uint32_t oe = o1 + (uint32_t) i*E1;
          const int pb0 = q ? (int) lb[q-1] : 0, pb1 = (int) lb[q];
          const uint64_t ks = (MODE == MODE_SELF) ? keyB[i] : lds_read_key(rawd,oe);
          int low, hgh, plen, lbnd, lnb, lna;       // lnb / lna: LCP with the T2 neighbour before / after (-1: none)
          if (MODE == MODE_SELF)
            { int lk  = (i > pb0)   ? lcp_key(ks,keyB[i-1]) : 0;
              int lk1 = (i+1 < pb1) ? lcp_key(ks,keyB[i+1]) : 11;
              plen = lk > lk1 ? lk : lk1;
              low = i; hgh = i+1; lbnd = i;
              lnb = (i > pb0) ? lk : -1; lna = (i+1 < pb1) ? lk1 : -1;
            }
          else
            { int lo = pb0, hi = pb1;
              const uint64_t kq = ks & ~0xffull;
              while (lo < hi)
                { int m = (lo+hi) >> 1;
                  if (keyB[m] < kq) lo = m+1; else hi = m;
                }
              int la_ = (lo > pb0) ? lcp_key(ks,keyB[lo-1]) : 0;
              int lc_ = (lo < pb1) ? lcp_key(ks,keyB[lo]) : 0;
              plen = la_ > lc_ ? la_ : lc_;
              low = hgh = lbnd = lo;
              lnb = (lo > pb0) ? la_ : -1; lna = (lo < pb1) ? lc_ : -1;
            }
          // run growth.  The first step on either side compares s with its T2 neighbour (lnb / lna, known already);
          // every further step asks whether the NEXT T2 entry still shares plen bases with its own predecessor,
          // which is the table's lcp byte (byte 8 of the entry, exact inside a panel -- the reference's vlcp[] walk,
          // FastGA.c:760-820, relies on the same bytes): one dword read instead of a 64-bit key compare
          if (lnb >= plen)
            { low -= 1;
              while (low > pb0 && lbnd-low <= freq && (int) T2_LCP(low) >= plen)
                low -= 1;
            }
          if (lna >= plen && hgh < pb1 && hgh-low <= freq)
            { hgh += 1;
              while (hgh < pb1 && hgh-low <= freq && (int) T2_LCP(hgh) >= plen)
                hgh += 1;
            }
          if (hgh-low >= freq)
            continue;
          const int mlen = A.soft_mask ? plen : 41;
          if ((int) (ks & 0xff) >= mlen)
            continue;
          int cnt;
          if (MODE == MODE_FLIP || A.soft_mask)
            { cnt = 0;
              for (int j = low; j < hgh; j++)
                { if ((int) (keyB[j] & 0xff) >= mlen)
                    continue;
                  if (MODE == MODE_FLIP)
                    { uint32_t sb = o2 + (uint32_t) j*E2 + E2 - 1;
                      if ((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80)
                        continue;
                    }
                  if (MODE == MODE_SELF && j == i)
                    continue;
                  cnt += 1;
                }
            }
          else
            cnt = (hgh-low) - (MODE == MODE_SELF ? 1 : 0);
          r_cnt[r] = cnt; r_low[r] = low | (hgh << 16); r_plen[r] = plen;
          total += cnt;
          tsum  += (unsigned long long) cnt * plen;
        }

#if defined(KNOCK_AFTER_MATCH)
      tsum += total;
      WSYNC();
      continue;
#endif
      // 5. slots: the lane's seeds take slots [off, off+total) of the wavefront's T, mapped onto the rest of the
      //    current chunk and, beyond it, a freshly reserved one
      int T;
      int off = wave_excl_scan_add_dpp(total,T);
      if (T > 0)
        { const int64_t rem = chunk_end - chunk_pos;
          int64_t nbase = 0, nsize = 0;
          if ((int64_t) T > rem)
            { nsize = ((int64_t) T - rem) > CHUNK_SEEDS ? ((int64_t) T - rem) : CHUNK_SEEDS;
              unsigned long long b = 0;
              if (lane == 0)
                b = atomicAdd(A.count,(unsigned long long) nsize);
              const uint32_t blo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) b);
              const uint32_t bhi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (b >> 32));
              nbase = (int64_t) (((uint64_t) bhi << 32) | blo);
            }
          // Emission.  Common case (no per-pair filter, at most WTILE_COST seeds in the tile): seed-parallel -- lane = output
          // slot.  Every entry with seeds leaves a one-dword descriptor (i, low, plen, first slot) in the compaction
          // list and marks its first slot with its rank; a wave max-scan over the slots finds each slot's entry, so the
          // wavefront runs ceil(T/64) full iterations instead of (rounds x longest run) mostly idle ones.
          const bool fast = (MODE != MODE_FLIP) && !A.soft_mask && T <= WTILE_COST;
          if (fast)
            { ((uint2 *) own)[lane] = make_uint2(0,0);
              WSYNC();
              { int o = off;
                #pragma unroll
                for (int r = 0; r < WEPT; r++)
                  if (r_cnt[r] > 0)
                    { clist[r*64 + lane] = (uint32_t) r_i[r] | ((uint32_t) (r_low[r] & 0xffff) << 8) |
                                           ((uint32_t) r_plen[r] << 16) | ((uint32_t) o << 22);
                      own[o] = (uint16_t) (lane*WEPT + r + 1);          // ranks grow with the slot number
                      o += r_cnt[r];
                    }
              }
              WSYNC();
              int carry = 0;
              for (int s0 = 0; s0 < T; s0 += 64)
                { const int slot = s0 + lane;
                  int v = slot < T ? (int) own[slot] : 0;
                  v = wave_incl_scan_max_dpp(v);
                  v = v > carry ? v : carry;
                  carry = __builtin_amdgcn_readlane(v,63);
                  if (slot < T)
                    { const int id = v-1;
                      const uint32_t d = clist[(id & (WEPT-1))*64 + (id >> 2)];
                      const int i = (int) (d & 0xff), plen = (int) ((d >> 16) & 0x3f);
                      int j = (int) ((d >> 8) & 0xff) + (slot - (int) (d >> 22));
                      if (MODE == MODE_SELF && j >= i)
                        j += 1;
                      uint32_t e0, e1_, e2_, e3, spos, sctg, ssign, c0, c1, c2, c3, cpos, cctg, csign;
                      lds_read16(rawd,o1 + (uint32_t) i*E1,e0,e1_,e2_,e3);
                      split_payload(e2_,e3,A.post1,A.cont1,spos,sctg,ssign);
                      lds_read16(rawd,o2 + (uint32_t) j*E2,c0,c1,c2,c3);
                      split_payload(c2,c3,A.post2,A.cont2,cpos,cctg,csign);
                      const int64_t at = ((int64_t) slot < rem) ? chunk_pos + slot : nbase + ((int64_t) slot - rem);
#if defined(KNOCK_STORE)
                      if (at == -12345)
#else
                      if (at < A.cap)
#endif
                        A.out[at] = make_seed<MODE>(plen,spos,sctg,ssign,cpos,cctg,csign);
                    }
                }
            }
          else
          if (total > 0)
            { const int mfull = A.soft_mask;
              #pragma unroll
              for (int r = 0; r < WEPT; r++)
                { if (r_cnt[r] == 0)
                    continue;
                  const int i = r_i[r];
                  const int low = r_low[r] & 0xffff, hgh = r_low[r] >> 16, plen = r_plen[r];
                  const int mlen = mfull ? plen : 41;
                  uint32_t e0, e1_, e2_, e3, spos, sctg, ssign;
                  lds_read16(rawd,o1 + (uint32_t) i*E1,e0,e1_,e2_,e3);
                  split_payload(e2_,e3,A.post1,A.cont1,spos,sctg,ssign);
                  for (int j = low; j < hgh; j++)
                    { if ((int) (keyB[j] & 0xff) >= mlen)
                        continue;
                      if (MODE == MODE_SELF && j == i)
                        continue;
                      uint32_t c0, c1, c2, c3, cpos, cctg, csign;
                      lds_read16(rawd,o2 + (uint32_t) j*E2,c0,c1,c2,c3);
                      split_payload(c2,c3,A.post2,A.cont2,cpos,cctg,csign);
                      if (MODE == MODE_FLIP && csign)
                        continue;
                      const int64_t at = ((int64_t) off < rem) ? chunk_pos + off : nbase + ((int64_t) off - rem);
#if defined(KNOCK_STORE)
                      if (at == -12345)
#else
                      if (at < A.cap)
#endif
                        A.out[at] = make_seed<MODE>(plen,spos,sctg,ssign,cpos,cctg,csign);
                      off += 1;
                    }
                }
            }
          if ((int64_t) T > rem)
            { chunk_pos = nbase + ((int64_t) T - rem); chunk_end = nbase + nsize; }
          else
            chunk_pos += T;
        }
      WSYNC();      // the tile buffers are reused by the next tile
    }

  if (lane == 0 && chunk_end > chunk_pos)          // the unused tail of the last chunk
    { const unsigned long long h = atomicAdd(W.ctr+0,1ull);
      if ((int64_t) h < W.hole_cap)
        { W.holes[2*h] = (unsigned long long) chunk_pos; W.holes[2*h+1] = (unsigned long long) chunk_end; }
    }
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0 && tsum != 0)
    atomicAdd(A.tseed,tsum);
}

// ---------------------------------------------------------------------------------------------------
// the range-walking wave kernel (v3)
// ---------------------------------------------------------------------------------------------------
// One launch does the whole merge.  A wavefront takes RANGES of consecutive 12-mer prefixes off a queue (equal merge
// cost per range, a few per wavefront) and walks each one tile by tile: it reads the next 64 entries of both prefix
// indices (one per lane, fetched one tile ahead), a ballot over the running cost finds how many prefixes fit a tile of
// XT cost units, and the tile is processed exactly as before -- raw bytes HBM -> LDS, T2 keys, panel owners,
// compaction, match, seed-parallel emission.  So there is no partition kernel and there are no tile descriptors.
// A single k-mer panel that exceeds a tile (a repeat family) is cut into sub-tiles IN the wavefront: a run of T1
// entries together with the stretch of the T2 panel its members can reach -- the lower bounds of its first and last
// key (a 64-ary search, the whole wavefront probing) widened by FREQ+2 entries, which is as far as the run growth of
// the merge ever looks; for a self comparison the stretch is the run itself plus that margin.  So there is no second
// kernel for oversize tiles either.
#ifndef XT
#define XT 256                       // cost units per tile
#endif
#define XEPT   (XT/64)               // T1 entries per lane in the owner / compaction pass; match rounds
#define XPC    64                    // prefixes per tile at most (one index entry per lane)
static_assert(XT == 256 || XT == 512,"descriptor packing: 9-bit tile indices, 4 or 8 entries per lane");

#ifndef WALK_PCOST
#define WALK_PCOST 1                 // cost units a prefix adds to a tile on top of its entries
#endif
#ifndef RANGES_PER_WAVE
#define RANGES_PER_WAVE 4
#endif
struct walk_args
  { const int64_t *cuts;             // [nranges+1] prefix boundaries
    int            nranges;
    int           *next;             // range queue head
  };

__global__ void range_cut_kernel(const int64_t *idx1, const int64_t *idx2, int pbeg, int pend, int64_t base,
                                 int64_t total, int nranges, int64_t *cuts)
{ const int w = blockIdx.x*blockDim.x + threadIdx.x;
  if (w > nranges)
    return;
  int64_t p = pbeg;
  if (w == nranges)
    p = pend;
  else if (w > 0)
    { const int64_t target = base + (total / nranges) * w;
      int lo = pbeg, hi = pend;
      while (lo < hi)
        { const int mid = lo + ((hi-lo) >> 1);
          const int64_t c = idx1[mid] + idx2[mid] + 2*((int64_t) mid+1);
          if (c > target) hi = mid; else lo = mid+1;
        }
      p = lo;
    }
  cuts[w] = p;
}

// key of table entry j straight from HBM: three aligned dwords, v_alignbyte, byte swap (the entry is >= 11 bytes)
__device__ __forceinline__ uint64_t glb_key3(const uint8_t *tab, int64_t j, int E)
{ const uint64_t addr = (uint64_t) (tab + j*E);
  const uint32_t *w = (const uint32_t *) (addr & ~(uint64_t) 3);
  const uint32_t sh = (uint32_t) (addr & 3);
  const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
  const uint32_t e0 = __builtin_amdgcn_alignbyte(d1,d0,sh), e1 = __builtin_amdgcn_alignbyte(d2,d1,sh);
  return ((uint64_t) bswap32(e0) << 32) | bswap32(e1);
}

// smallest j in [lo,hi] whose key is >= kq (mask byte ignored), all 64 lanes probing: the range shrinks 64-fold a round
__device__ __forceinline__ int64_t wave_lower_bound(const uint8_t *tab, int E, int64_t lo, int64_t hi, uint64_t kq)
{ const int lane = threadIdx.x;
  while (hi > lo)
    { const int64_t step = ((hi - lo) + 63) >> 6;
      const int64_t pos = lo + (int64_t) lane*step;
      bool below = false;
      if (pos < hi)
        below = (glb_key3(tab,pos,E) & ~0xffull) < kq;
      const int t = __popcll(__builtin_amdgcn_ballot_w64(below));        // sorted: the first t probes are below
      if (t == 0)
        return lo;
      const int64_t nhi = lo + (int64_t) t*step;
      lo = lo + (int64_t) (t-1)*step + 1;
      if (nhi < hi) hi = nhi;
    }
  return lo;
}

// key and lcp byte (byte 8) of the entry at byte offset o of the staged bytes
__device__ __forceinline__ uint64_t lds_read_key_lcp(const uint32_t *rawd, uint32_t o, uint32_t &lcp)
{ const uint32_t w = o >> 2, sh = o & 3;
  const uint32_t d0 = rawd[w], d1 = rawd[w+1], d2 = rawd[w+2];
  const uint32_t e0 = __builtin_amdgcn_alignbyte(d1,d0,sh);
  const uint32_t e1 = __builtin_amdgcn_alignbyte(d2,d1,sh);
  lcp = (d2 >> (8*sh)) & 0xff;
  return ((uint64_t) bswap32(e0) << 32) | bswap32(e1);
}

// payload (position, contig, sign) of the entry at byte offset o of the staged bytes: bytes 9.. of it, three dwords
__device__ __forceinline__ void lds_payload(const uint32_t *rawd, uint32_t o, int post, int cont,
                                            uint32_t &pos, uint32_t &ctg, uint32_t &sign)
{ const uint32_t w = (o + 9) >> 2, sh = (o + 9) & 3;
  const uint32_t d0 = rawd[w], d1 = rawd[w+1], d2 = rawd[w+2];
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1,d0,sh), hi = __builtin_amdgcn_alignbyte(d2,d1,sh);
  const uint64_t pv = ((uint64_t) hi << 32) | lo;
  const uint32_t pm = post >= 4 ? 0xffffffffu : ((1u << (8*post)) - 1);
  pos = lo & pm;
  const uint32_t c  = (uint32_t) (pv >> (8*post)) & ((1u << (8*cont)) - 1);
  const uint32_t sb = 0x80u << (8*(cont-1));
  sign = (c & sb) != 0;
  ctg  = c & (sb-1);
}

#ifdef MERGE_PROF
#define XPROF(k)   { unsigned long long _n = clock64(); O.pa[k] += _n - O.pt; O.pt = _n; }
#else
#define XPROF(k)
#endif
struct walk_out                      // a wavefront's current output chunk and statistics (wave-uniform)
  { int64_t chunk_pos, chunk_end;
    unsigned long long tsum;
#ifdef MERGE_PROF
    unsigned long long pt, pa[8];
#endif
  };

// One tile: T1 entries [a0, a0+n1) and T2 entries [b0, b0+n2) (self: the same stretch) in np <= 64 panels whose
// cumulative sizes are already in la[] / lb[]; only the T1 entries [t1_lo, t1_hi) of the stretch emit.
template <int MODE>
__device__ __forceinline__ void walk_tile(const merge_args &A, uint16_t *la, uint16_t *lb, uint8_t *raw, uint64_t *keyB,
                                          uint16_t *own, uint8_t *lcpB, int64_t a0, int n1, int64_t b0, int n2, int np,
                                          int t1_lo, int t1_hi, walk_out &O)
{ const int lane = threadIdx.x;
  const int E1 = A.E1, E2 = A.E2;
  const int freq = A.freq;
  uint32_t *own32 = (uint32_t *) keyB;           // the seed-parallel emission reuses the key array (keys are done with by then)
  const int64_t s1 = a0*E1, e1 = (a0+n1)*E1;
  const int64_t s2 = b0*E2, e2 = (b0+n2)*E2;
  const int64_t s1a = s1 & ~(int64_t) 15, s2a = s2 & ~(int64_t) 15;
  const int64_t len1 = ((e1 - s1a) + 15) & ~(int64_t) 15;
  const int64_t len2 = (MODE == MODE_SELF) ? 0 : (((e2 - s2a) + 15) & ~(int64_t) 15);

  // 1. raw bytes HBM -> LDS (global_load_lds_dwordx4), head-flag array cleared
  { const uint4 *g1 = (const uint4 *) (A.tab1 + s1a);
    uint4 *l1 = (uint4 *) raw;
    const int n16 = (int) (len1 >> 4);
    for (int x = lane; x < n16; x += 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g1 + x),
                                       (__attribute__((address_space(3))) void *) (l1 + x),16,0,0);
    if (MODE != MODE_SELF)
      { const uint4 *g2 = (const uint4 *) (A.tab2 + s2a);
        uint4 *l2 = (uint4 *) (raw + len1);
        const int m16 = (int) (len2 >> 4);
        for (int x = lane; x < m16; x += 64)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g2 + x),
                                           (__attribute__((address_space(3))) void *) (l2 + x),16,0,0);
      }
  }
  #pragma unroll
  for (int x = 0; x < XT/256; x++)
    ((uint2 *) own)[x*64 + lane] = make_uint2(0,0);
  // the direct-to-LDS loads are asynchronous and nothing below depends on a VGPR they return: wait for them by hand
  // (vmcnt(0); the counters of the other queues are left alone)
  XPROF(0)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  WSYNC();
  XPROF(1)
#if defined(XKNOCK) && XKNOCK == 1                   // phase knock-outs: timing experiments only (wrong output)
  return;
#endif

  const uint32_t *rawd = (const uint32_t *) raw;
  const uint32_t o1 = (uint32_t) (s1 - s1a);
  const uint32_t o2 = (MODE == MODE_SELF) ? o1 : (uint32_t) (len1 + (s2 - s2a));

  // 2. head flags of the non-empty T1 panels; T2 keys
  if (lane < np)
    { const uint32_t s = lane ? la[lane-1] : 0;
      if (la[lane] > s)
        own[s] = (uint16_t) lane;
    }
  for (int j = lane; j < n2; j += 64)
    { uint32_t lc;
      keyB[j] = lds_read_key_lcp(rawd,o2 + (uint32_t) j*E2,lc);
      lcpB[j] = (uint8_t) lc;                          // the run growth below walks these bytes
    }
  WSYNC();
  XPROF(2)
#if defined(XKNOCK) && XKNOCK == 2
  return;
#endif

  // 3. owner of every T1 entry (XEPT consecutive entries per lane + wave max-scan); compaction of those that can emit
  int nlive;
  uint32_t *clist = (uint32_t *) (keyB + n2);
  { int xs[XEPT];
    { const uint2 v = ((const uint2 *) own)[lane*(XEPT/4)];
      xs[0] = v.x & 0xffff; xs[1] = v.x >> 16; xs[2] = v.y & 0xffff; xs[3] = v.y >> 16;
    }
    if (XEPT == 8)
      { const uint2 v = ((const uint2 *) own)[lane*2+1];
        xs[XEPT-4] = v.x & 0xffff; xs[XEPT-3] = v.x >> 16; xs[XEPT-2] = v.y & 0xffff; xs[XEPT-1] = v.y >> 16;
      }
    #pragma unroll
    for (int e = 1; e < XEPT; e++)
      xs[e] = xs[e] > xs[e-1] ? xs[e] : xs[e-1];
    const int inc = wave_incl_scan_max_dpp(xs[XEPT-1]);
    const int prev = __builtin_amdgcn_update_dpp(0,inc,0x138,0xf,0xf,false);      // lane-1's inclusive value, 0 for lane 0
    int live = 0;
    uint32_t pk[XEPT];
    #pragma unroll
    for (int e = 0; e < XEPT; e++)
      { const int i = lane*XEPT + e;
        const int q = xs[e] > prev ? xs[e] : prev;
        bool ok = i >= t1_lo && i < t1_hi;
        int pb0 = 0, pb1 = 0;
        if (ok)
          { pb0 = q ? (int) lb[q-1] : 0; pb1 = (int) lb[q];
            ok = pb1 > pb0;
            if (ok && MODE == MODE_PAIR)
              { const uint32_t sb = o1 + (uint32_t) i*E1 + E1 - 1;
                ok = !((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80);
              }
          }
        // the panel bounds travel with the entry: the match needs neither the owner nor lb[] again
        pk[e] = ok ? ((uint32_t) i | ((uint32_t) pb0 << 10) | ((uint32_t) pb1 << 20)) : 0xffffffffu;
        live += ok;
      }
    int lbase = wave_excl_scan_add_dpp(live,nlive);
    #pragma unroll
    for (int e = 0; e < XEPT; e++)
      if (pk[e] != 0xffffffffu)
        clist[lbase++] = pk[e];
  }
  WSYNC();
  XPROF(3)
#if defined(XKNOCK) && XKNOCK == 3
  O.tsum += nlive;
  return;
#endif

  // 4. match phase; result per round packed: i (9 bits) | low << 9 | plen << 18 | seeds << 24.
  // Written without branches around its LDS reads, so that the reads of one step are in flight together: the entry,
  // then its key, then the lower bound of the key in its T2 panel (a wave-uniform number of halving steps), then both
  // neighbour keys and the first lcp byte of either growth direction in one round trip.  Only runs that grow past
  // their first step (repeats) take a loop.
  uint32_t res[XEPT];
  int total = 0;
  #pragma unroll
  for (int r = 0; r < XEPT; r++)
    { res[r] = 0;
      if (r*64 >= nlive)                       // wave-uniform
        continue;
      const int c = r*64 + lane;
      const bool act = c < nlive;
      const uint32_t ce = act ? clist[c] : 0u;
      const int i = (int) (ce & 0x3ff), pb0 = (int) ((ce >> 10) & 0x3ff), pb1 = (int) (ce >> 20);
      const uint64_t ks = (MODE == MODE_SELF) ? keyB[i] : lds_read_key(rawd,o1 + (uint32_t) i*E1);
      int nb, na, low, hgh, lbnd;              // nb / na: the T2 neighbour before / after
      if (MODE == MODE_SELF)
        { nb = i-1; na = i+1; low = i; hgh = i+1; lbnd = i; }
      else
        { const uint64_t kq = ks & ~0xffull;
          int base = pb0, len = pb1 - pb0;     // first entry of [pb0,pb1) that is >= kq
          while (__builtin_amdgcn_ballot_w64(len > 1) != 0)
            { const int half = len >> 1;
              const bool lt = keyB[base + half - 1] < kq && len > 1;
              base += lt ? half : 0;
              len  -= (len > 1) ? half : 0;
            }
          base += (len > 0 && keyB[base] < kq) ? 1 : 0;
          nb = base-1; na = base; low = hgh = lbnd = base;
        }
      const uint64_t kb = keyB[nb], kc = keyB[na];       // keyB[-1] and keyB[n2] exist (never used when out of the panel)
      const int bd = (int) lcpB[nb], bu = (int) lcpB[na+1];
      const bool hasb = nb >= pb0, hasa = na < pb1;
      const int lkb = lcp_key(ks,kb), lka = lcp_key(ks,kc);
      const int lnb = hasb ? lkb : -1, lna = hasa ? lka : -1;       // -1: no such neighbour
      const int pb_ = hasb ? lkb : 0, pa_ = hasa ? lka : (MODE == MODE_SELF ? 11 : 0);
      const int plen = pb_ > pa_ ? pb_ : pa_;
      // run growth on the table's own lcp bytes (see the wave kernel above); the first step of either direction is
      // decided on the bytes read above
      const bool gd = act && lnb >= plen;
      low -= gd ? 1 : 0;
      if (gd && low > pb0 && lbnd-low <= freq && bd >= plen)
        { low -= 1;
          while (low > pb0 && lbnd-low <= freq && (int) lcpB[low] >= plen)
            low -= 1;
        }
      const bool gu = act && lna >= plen && hgh < pb1 && hgh-low <= freq;
      hgh += gu ? 1 : 0;
      if (gu && hgh < pb1 && hgh-low <= freq && bu >= plen)
        { hgh += 1;
          while (hgh < pb1 && hgh-low <= freq && (int) lcpB[hgh] >= plen)
            hgh += 1;
        }
      const int mlen = A.soft_mask ? plen : 41;
      const bool pass = act && hgh-low < freq && (int) (ks & 0xff) < mlen;
      int cnt;
      if (MODE == MODE_FLIP || A.soft_mask)
        { cnt = 0;
          if (pass)
            for (int j = low; j < hgh; j++)
              { if ((int) (keyB[j] & 0xff) >= mlen)
                  continue;
                if (MODE == MODE_FLIP)
                  { uint32_t sb = o2 + (uint32_t) j*E2 + E2 - 1;
                    if ((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80)
                      continue;
                  }
                if (MODE == MODE_SELF && j == i)
                  continue;
                cnt += 1;
              }
        }
      else
        cnt = pass ? (hgh-low) - (MODE == MODE_SELF ? 1 : 0) : 0;
      res[r] = pass ? ((uint32_t) i | ((uint32_t) low << 9) | ((uint32_t) plen << 18) | ((uint32_t) cnt << 24)) : 0u;
      total += cnt;
      O.tsum += (unsigned long long) cnt * plen;
    }

  XPROF(4)
#if defined(XKNOCK) && XKNOCK == 4
  O.tsum += total;
  WSYNC();
  return;
#endif
  // 5. slots and emission
  int T;
  int off = wave_excl_scan_add_dpp(total,T);
  if (T > 0)
    { const int64_t rem = O.chunk_end - O.chunk_pos;
      int64_t nbase = 0, nsize = 0;
      if ((int64_t) T > rem)
        { nsize = (((int64_t) T - rem) + FGA_SEED_BLOCK-1) & ~(int64_t) (FGA_SEED_BLOCK-1);    // whole blocks, block aligned
          unsigned long long b = 0;
          if (lane == 0)
            b = atomicAdd(A.count,(unsigned long long) nsize);
          const uint32_t blo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) b);
          const uint32_t bhi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (b >> 32));
          nbase = (int64_t) (((uint64_t) bhi << 32) | blo);
        }
      // Seed-parallel, in windows of 2 XT slots (the key array's size in dwords): every entry with seeds in the window
      // leaves a descriptor at its first slot there -- i | low << 9 | plen << 18 | (seeds of its run before the window) << 24
      // -- and a wave max-scan over "slot+1 where a descriptor sits" tells every slot where its entry's run begins.
      // The lane of a slot then finds its partner: the k-th T2 entry of the run, or, where mask bytes / strands / the
      // entry itself drop members of the run, the k-th one that stays.
      const bool plain = (MODE != MODE_FLIP) && !A.soft_mask;
      for (int wb = 0; wb < T; wb += 2*XT)
        { const int wn = T - wb < 2*XT ? T - wb : 2*XT;          // slots of this window
          for (int x = lane; 4*x < wn; x += 64)
            ((uint4 *) own32)[x] = make_uint4(0,0,0,0);
          WSYNC();
          { int o = off;
            #pragma unroll
            for (int r = 0; r < XEPT; r++)
              { const int cnt = (int) (res[r] >> 24);
                if (cnt > 0 && o + cnt > wb && o < wb + 2*XT)
                  { const int before = o < wb ? wb - o : 0;
                    own32[o + before - wb] = (res[r] & 0xffffffu) | ((uint32_t) before << 24) | 0x80000000u;
                  }
                o += cnt;
              }
          }
          WSYNC();
          int carry = 0;
          for (int s0 = 0; s0 < wn; s0 += 64)
            { const int slot = s0 + lane;                       // within the window
              int v = (slot < wn && own32[slot] != 0) ? slot+1 : 0;
              v = wave_incl_scan_max_dpp(v);
              v = v > carry ? v : carry;
              carry = __builtin_amdgcn_readlane(v,63);
              if (slot < wn)
                { const int start = v-1;
                  const uint32_t d = own32[start];
                  const int i = (int) (d & 0x1ff), plen = (int) ((d >> 18) & 0x3f);
                  int k = (slot - start) + (int) ((d >> 24) & 0x7f);
                  int j = (int) ((d >> 9) & 0x1ff);
                  if (plain)
                    { j += k;
                      if (MODE == MODE_SELF && j >= i)
                        j += 1;
                    }
                  else
                    { const int mlen = A.soft_mask ? plen : 41;
                      for (;; j++)
                        { const uint32_t mb = o2 + (uint32_t) j*E2 + 7;          // the entry's mask byte
                          if ((int) ((rawd[mb >> 2] >> (8*(mb & 3))) & 0xff) >= mlen)
                            continue;
                          if (MODE == MODE_FLIP)
                            { const uint32_t sb = o2 + (uint32_t) j*E2 + E2 - 1;
                              if ((rawd[sb >> 2] >> (8*(sb & 3))) & 0x80)
                                continue;
                            }
                          if (MODE == MODE_SELF && j == i)
                            continue;
                          if (k == 0)
                            break;
                          k -= 1;
                        }
                    }
                  uint32_t spos, sctg, ssign, cpos, cctg, csign;
                  lds_payload(rawd,o1 + (uint32_t) i*E1,A.post1,A.cont1,spos,sctg,ssign);
                  lds_payload(rawd,o2 + (uint32_t) j*E2,A.post2,A.cont2,cpos,cctg,csign);
                  const int64_t gs = (int64_t) wb + slot;
                  const int64_t at = (gs < rem) ? O.chunk_pos + gs : nbase + (gs - rem);
#if defined(XKNOCK) && XKNOCK == 5
                  if (at == -12345)
#else
                  if (at < A.cap)
#endif
                    A.out[at] = make_seed<MODE>(plen,spos,sctg,ssign,cpos,cctg,csign);
                }
            }
          WSYNC();
        }
      if ((int64_t) T > rem)
        { O.chunk_pos = nbase + ((int64_t) T - rem); O.chunk_end = nbase + nsize; }
      else
        O.chunk_pos += T;
    }
  WSYNC();      // the tile buffers are reused by the next tile
  XPROF(5)
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVE_OCC,WAVE_OCC)))
void seed_merge_walk_kernel(merge_args A, walk_args W)
{ __shared__ uint16_t la[XPC+1];
  __shared__ uint16_t lb[XPC+1];
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];      // A.wrawcap bytes (sized by the entry widths)
  __shared__ __attribute__((aligned(16))) uint64_t keyB0[XT+4];      // keyB[-2 .. XT+1]: the match reads one past either end
  __shared__ __attribute__((aligned(16))) uint16_t own16[XT];
  __shared__ __attribute__((aligned(16))) uint8_t  lcpB0[XT+16];     // lcpB[-8 .. XT+7]
  __shared__ __attribute__((aligned(16))) uint32_t ixs[4][64];       // index entries of the next 64 prefixes (lo / hi words)
  uint64_t *keyB = keyB0 + 2;
  uint8_t  *lcpB = lcpB0 + 8;

  const int lane = threadIdx.x;
  const int E1 = A.E1, E2 = A.E2;
  const int64_t *idx2 = (MODE == MODE_SELF) ? A.idx1 : A.idx2;
  const uint8_t *tab2 = (MODE == MODE_SELF) ? A.tab1 : A.tab2;
  const int margin = A.freq + 2;
  walk_out O;
  O.chunk_pos = O.chunk_end = 0; O.tsum = 0;
#ifdef MERGE_PROF
  O.pt = clock64();
  for (int k = 0; k < 8; k++) O.pa[k] = 0;
#endif

  for (;;)
    { int r = 0;
      if (lane == 0)
        r = atomicAdd(W.next,1);
      r = __builtin_amdgcn_readfirstlane(r);
      if (r >= W.nranges)
        break;
      int p = (int) W.cuts[r];
      const int pe = (int) W.cuts[r+1];
      if (p >= pe)
        continue;
      int64_t a = idx_at(A.idx1,p-1), b = idx_at(idx2,p-1);
      // Index entries of the next 64 prefixes, one per lane (clamped at the range end).  They are fetched one tile
      // ahead and travel HBM -> LDS like the tile bytes, not into registers: a register result would make the compiler
      // wait for ALL outstanding vector-memory operations where the loop uses it, i.e. for the acknowledgements of the
      // seed stores the tile before has just issued.  Their arrival is covered by the tile's own wait for its bytes
      // (issued after them, completed in order); the reads below are opaque to the compiler for the same reason.
#define IDX_ISSUE(P0)                                                                                                \
      { const int64_t e_ = (P0) + (lane < pe-(P0) ? lane : pe-(P0)-1);                                                  \
        const uint32_t *g_ = (const uint32_t *) (A.idx1 + e_);                                                         \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g_,                          \
                                         (__attribute__((address_space(3))) void *) &ixs[0][lane],4,0,0);               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (g_+1),                      \
                                         (__attribute__((address_space(3))) void *) &ixs[1][lane],4,0,0);               \
        if (MODE != MODE_SELF)                                                                                         \
          { const uint32_t *h_ = (const uint32_t *) (idx2 + e_);                                                       \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) h_,                      \
                                             (__attribute__((address_space(3))) void *) &ixs[2][lane],4,0,0);           \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (h_+1),                  \
                                             (__attribute__((address_space(3))) void *) &ixs[3][lane],4,0,0);           \
          }                                                                                                            \
      }
      IDX_ISSUE(p)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      while (p < pe)
        { int64_t ca, cb;
          { const uint32_t at = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) &ixs[0][lane];
            uint64_t va, vb;
            asm volatile("ds_read2st64_b32 %0, %2 offset1:1\n\tds_read2st64_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(va), "=&v"(vb) : "v"(at) : "memory");
            ca = (int64_t) va;
            cb = (MODE == MODE_SELF) ? ca : (int64_t) vb;
          }
          const int navail = pe - p < XPC ? pe - p : XPC;
          const int64_t cost = (ca - a) + (cb - b) + WALK_PCOST*((int64_t) lane+1);
          const bool fits = lane < navail && cost <= XT;
          const int q = __popcll(__builtin_amdgcn_ballot_w64(fits));     // cost grows with the lane: a prefix mask
          XPROF(6)
          const int adv = q > 0 ? q : 1;
          // end of the tile (or of the single oversize panel): totals up to prefix p+adv-1
          const uint32_t alo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) ca,adv-1);
          const uint32_t ahi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) ((uint64_t) ca >> 32),adv-1);
          const uint32_t blo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) cb,adv-1);
          const uint32_t bhi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) ((uint64_t) cb >> 32),adv-1);
          const int64_t a1 = (int64_t) (((uint64_t) ahi << 32) | alo), b1 = (int64_t) (((uint64_t) bhi << 32) | blo);
          const int64_t n1 = a1 - a, n2 = b1 - b;
          if (q > 0 && n1 > 0 && n2 > 0 && lane < q)
            { la[lane] = (uint16_t) (ca - a);
              lb[lane] = (uint16_t) (cb - b);
            }
          // the index entries of the tile after this one are on their way while this one is processed
          const int pn = p + adv;
          if (pn < pe)
            IDX_ISSUE(pn)
          XPROF(7)
          if (n1 > 0 && n2 > 0)
            { if (q > 0)
                walk_tile<MODE>(A,la,lb,raw,keyB,own16,lcpB,a,(int) n1,b,(int) n2,q,0,(int) n1,O);
#ifdef SKIP_BIG                      // timing experiments only (wrong output): oversize panels dropped
              else if (true) ;
#endif
              else if (MODE == MODE_SELF)
                { // one oversize panel against itself: runs of T1 entries with `margin` neighbours either side
                  const int C = XT/2 - 2*margin;
                  for (int64_t i0 = 0; i0 < n1; i0 += C)
                    { const int64_t i1 = i0 + C < n1 ? i0 + C : n1;
                      const int64_t s0 = i0 - margin > 0 ? i0 - margin : 0, s1 = i1 + margin < n1 ? i1 + margin : n1;
                      if (lane == 0)
                        la[0] = lb[0] = (uint16_t) (s1 - s0);
                      walk_tile<MODE>(A,la,lb,raw,keyB,own16,lcpB,a+s0,(int) (s1-s0),a+s0,(int) (s1-s0),1,
                                      (int) (i0-s0),(int) (i1-s0),O);
                    }
                }
              else
                { // one oversize panel: runs of T1 entries, each with the stretch of the T2 panel its keys can reach
                  int64_t from = b;
                  for (int64_t i = 0; i < n1; )
                    { int c1 = (int) (n1 - i < XT/2 ? n1 - i : XT/2);
                      int64_t s0, s1, l1;
                      const uint64_t k0 = glb_key3(A.tab1,a+i,E1) & ~0xffull;
                      const int64_t l0 = wave_lower_bound(tab2,E2,from,b1,k0);
                      for (;;)
                        { const uint64_t k1 = glb_key3(A.tab1,a+i+c1-1,E1) & ~0xffull;
                          l1 = wave_lower_bound(tab2,E2,l0,b1,k1);
                          s0 = l0 - margin > b ? l0 - margin : b;
                          s1 = l1 + margin < b1 ? l1 + margin : b1;
                          if (c1 + (s1 - s0) <= XT-2 || c1 == 1)
                            break;
                          c1 = c1 > 1 ? c1/2 : 1;
                        }
                      if (s1 > s0)
                        { if (lane == 0)
                            { la[0] = (uint16_t) c1; lb[0] = (uint16_t) (s1 - s0); }
                          walk_tile<MODE>(A,la,lb,raw,keyB,own16,lcpB,a+i,c1,s0,(int) (s1-s0),1,0,c1,O);
                        }
                      from = l1;
                      i += c1;
                    }
                }
            }
          if (!(q > 0 && n1 > 0 && n2 > 0))              // no tile, no wait of a tile: the index entries may be on their way
            __builtin_amdgcn_s_waitcnt(0x0F70);
          a = a1; b = b1; p = pn;
        }
#undef IDX_ISSUE
    }

  if (O.chunk_end > O.chunk_pos)                       // the unused tail of the last chunk stays open: its blocks say so
    { const int64_t b0 = O.chunk_pos >> 10, b1 = (O.chunk_end - 1) >> 10;
      for (int64_t bk = b0 + lane; bk <= b1; bk += 64)
        if (bk < A.nblocks)
          A.valid[bk] = (uint16_t) (bk == b0 ? (O.chunk_pos & (FGA_SEED_BLOCK-1)) : 0);
      if (lane == 0)
        atomicAdd(A.hslots,(unsigned long long) (O.chunk_end - O.chunk_pos));
    }
#ifdef MERGE_PROF
  if (lane == 0)
    for (int k = 0; k < 8; k++) atomicAdd(merge_prof+k,O.pa[k]);
#endif
  unsigned long long tsum = O.tsum;
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0 && tsum != 0)
    atomicAdd(A.tseed,tsum);
}

// tile descriptor pairs of the queued oversize tiles, for the workgroup kernel in pair mode
__global__ void gather_big_tiles_kernel(const merge_tile *tiles, const int *bigq, const unsigned long long *nbig,
                                        int big_cap, merge_tile *pairs)
{ const int k = blockIdx.x*blockDim.x + threadIdx.x;
  const int n = (int) (*nbig < (unsigned long long) big_cap ? *nbig : (unsigned long long) big_cap);
  if (k >= n)
    return;
  pairs[2*k]   = tiles[bigq[k]];
  pairs[2*k+1] = tiles[bigq[k]+1];
}

// close the holes: copy `len` seeds from src to dst for every planned move (one workgroup per move)
struct seed_move { int64_t src, dst, len; };
__global__ void hole_fill_kernel(fga_seed *seeds, const seed_move *moves, int nmoves)
{ if ((int) blockIdx.x >= nmoves)
    return;
  const seed_move m = moves[blockIdx.x];
  for (int64_t x = threadIdx.x; x < m.len; x += blockDim.x)
    seeds[m.dst + x] = seeds[m.src + x];
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static int merge_impl(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                      const fga_merge_params *prm, int64_t capacity, fga_dseeds **out, fga_dseeds *append)
{ if (out != NULL) *out = NULL;
  if (dev == NULL || t1 == NULL || prm == NULL)
    { fga_set_error("fga_seed_merge: null argument");
      return 1;
    }
  const int self = (t2 == NULL);
  if (self) t2 = t1;
  if (t1->ebytes > 16 || t2->ebytes > 16)
    { fga_set_error("fga_seed_merge: entries wider than 16 bytes are not supported");
      return 1;
    }
  if (self && prm->flip)
    { fga_set_error("fga_seed_merge: flip is meaningless for a self comparison");
      return 1;
    }
  FGA_HIP(hipSetDevice(dev->device));

  merge_args A;
  A.tab1 = t1->table; A.idx1 = t1->index; A.E1 = t1->ebytes; A.post1 = t1->postbytes; A.cont1 = t1->contbytes;
  A.tab2 = t2->table; A.idx2 = t2->index; A.E2 = t2->ebytes; A.post2 = t2->postbytes; A.cont2 = t2->contbytes;
  A.freq = prm->freq; A.soft_mask = prm->soft_mask; A.flip = prm->flip; A.self = self;
  A.pbeg = (int) prm->prefix_begin;
  A.pend = (int) prm->prefix_end;
  if (A.pend <= 0 || A.pend > FGA_NPREFIX) A.pend = FGA_NPREFIX;
  if (A.pbeg < 0) A.pbeg = 0;
  if (A.pbeg >= A.pend)
    { fga_set_error("fga_seed_merge: empty prefix range");
      return 1;
    }
  if (A.freq < 1 || A.freq > 255)
    { fga_set_error("fga_seed_merge: frequency cutoff must be in [1,255]");
      return 1;
    }

  // cost at both ends of the prefix range (4 tiny D2H copies)
  int64_t c1e, c2e, c1b = 0, c2b = 0;
  FGA_HIP(hipMemcpy(&c1e,t1->index + (A.pend-1),8,hipMemcpyDeviceToHost));
  FGA_HIP(hipMemcpy(&c2e,t2->index + (A.pend-1),8,hipMemcpyDeviceToHost));
  if (A.pbeg > 0)
    { FGA_HIP(hipMemcpy(&c1b,t1->index + (A.pbeg-1),8,hipMemcpyDeviceToHost));
      FGA_HIP(hipMemcpy(&c2b,t2->index + (A.pbeg-1),8,hipMemcpyDeviceToHost));
    }
  A.base = c1b + c2b + 2*(int64_t) A.pbeg;
  int64_t total = (c1e + c2e + 2*(int64_t) A.pend) - A.base;
  // FGA_MERGE_V1=1 selects the workgroup-per-tile kernel for everything (the wave kernel's fallback path otherwise)
  int use_wave = 1;
  { const char *e = getenv("FGA_MERGE_V1");
    if (e != NULL && atoi(e) != 0) use_wave = 0;
  }

  // the range-walking kernel (v3) does the whole merge in one launch; FGA_MERGE_V2=1 selects the previous wave kernel
  // (partition kernel + tile queue + workgroup kernel for oversize tiles), which also takes over when the frequency
  // cutoff is too large for the in-wavefront sub-tiles of an oversize panel (margin FREQ+2 on both sides)
  int use_walk = use_wave;
  { const char *e = getenv("FGA_MERGE_V2");
    if (e != NULL && atoi(e) != 0) use_walk = 0;
    if (prm->freq + 2 >= XT/4 - 1) use_walk = 0;
  }

  fga_dseeds *S = append;
  if (S == NULL)
    { S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
      if (S == NULL)
        { fga_set_error("out of memory");
          return 1;
        }
      S->dev = dev;
      if (capacity <= 0)
        capacity = 2*(c1e - c1b) + (1<<20);
      S->capacity = capacity;
      // every wavefront of the wave kernel may leave up to a chunk unused until the holes are closed, in this
      // call and in a later append (-S)
      S->phys_capacity = capacity + 2*(int64_t) dev->ncu * 32 * CHUNK_SEEDS;
    }
  else
    capacity = S->capacity;
  const int64_t phys = S->phys_capacity;

  unsigned long long *counters = NULL;
  hipError_t err;
  if (append != NULL)
    counters = (unsigned long long *) S->dcount;
  else
    { if ((err = hipMalloc(&counters,4*sizeof(unsigned long long))) != hipSuccess ||
          (S->seeds = (fga_seed *) fga_dev_acquire(dev,SLOT_SEEDS,sizeof(fga_seed)*(size_t) S->phys_capacity)) == NULL)
        { fga_set_error("fga_seed_merge: device allocation failed: %s",hipGetErrorString(err));
          hipFree(counters); free(S);
          return 1;
        }
      S->slot = SLOT_SEEDS;
      S->dcount = (int64_t *) counters;
      hipMemsetAsync(counters,0,4*sizeof(unsigned long long),dev->stream);
      if (use_walk)
        { const int64_t nb = (S->phys_capacity + FGA_SEED_BLOCK-1) / FGA_SEED_BLOCK + 2;
          S->valid = (uint16_t *) fga_dev_acquire(dev,SLOT_VALID,sizeof(uint16_t)*(size_t) nb);
          if (S->valid == NULL)
            { fga_set_error("fga_seed_merge: device allocation failed");
              hipFree(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); free(S);
              return 1;
            }
          hipMemsetD16Async((hipDeviceptr_t) S->valid,(unsigned short) FGA_SEED_BLOCK,(size_t) nb,dev->stream);
        }
    }
  if (S->valid == NULL) use_walk = 0;            // appending to a dense buffer of the previous kernels
  A.out = S->seeds; A.cap = phys;
  A.count = counters; A.tseed = counters+1; A.hslots = counters+2;
  A.valid = S->valid; A.nblocks = (S->phys_capacity + FGA_SEED_BLOCK-1) / FGA_SEED_BLOCK + 2;
  A.pairs = 0; A.npairs = NULL; A.pair_cap = 0;
  { const int emax = A.E1 > A.E2 ? A.E1 : A.E2;           // two 16-byte-aligned ranges of <= tile-cost entries in all
    A.wrawcap = (((use_walk ? XT : WTILE_COST)*emax + 96 + 15) / 16) * 16;
  }

  void *work = NULL;                      // tiles + (wave kernel) pairs, holes, counters, queue, moves
  unsigned long long hc[2];
  int64_t hslots = 0;                     // v3: slots the launch (and earlier ones into the same buffer) left unused
  int rc = 1;
  hipEvent_t ev2 = NULL;
  hipEventCreate(&ev2);

  for (int attempt = 0; attempt < 2; attempt++)
    { A.tile_cost = use_wave ? WTILE_COST : TILE_COST;
      A.ntiles = (int) (total / A.tile_cost) + 1;
      int wgs = use_wave ? 4*WAVE_OCC : 4;
      { const char *ev = getenv(use_wave ? "FGA_MERGE_WAVES" : "FGA_MERGE_WGS");
        if (ev != NULL && atoi(ev) > 0) wgs = atoi(ev);
      }
      if (use_walk) wgs = 28;
      if (use_wave && wgs > 28) wgs = 28;                    // the slack of phys_capacity covers 32 waves per CU
      int grid = dev->ncu * wgs;
      if (use_wave && grid > A.ntiles/8 + 1) grid = A.ntiles/8 + 1;      // small inputs: few waves, few holes
      if (grid > A.ntiles) grid = A.ntiles;
      const int big_cap  = use_wave ? (A.ntiles < (1<<18) ? A.ntiles : (1<<18)) : 0;
      const int hole_cap = use_wave ? grid + 16 : 0;
      const size_t tile_bytes = sizeof(merge_tile)*(size_t) (A.ntiles+1);
      const size_t pair_bytes = sizeof(merge_tile)*2*(size_t) big_cap;
      const size_t hole_bytes = sizeof(unsigned long long)*2*(size_t) hole_cap;
      const size_t move_bytes = sizeof(seed_move)*2*(size_t) (hole_cap+1);
      const size_t q_bytes    = sizeof(int)*(size_t) big_cap;
      work = fga_dev_acquire(dev,SLOT_TILES,tile_bytes + pair_bytes + hole_bytes + move_bytes + q_bytes + 256);
      if (work == NULL)
        { fga_set_error("fga_seed_merge: device allocation failed");
          goto done;
        }
      merge_tile *tiles = (merge_tile *) work;
      merge_tile *pairs = tiles + (A.ntiles+1);
      unsigned long long *holes = (unsigned long long *) (pairs + 2*(size_t) big_cap);
      seed_move *moves = (seed_move *) (holes + 2*(size_t) hole_cap);
      unsigned long long *wctr = (unsigned long long *) (moves + 2*(size_t) (hole_cap+1));
      int *bigq = (int *) (wctr + 4);
      A.tiles = tiles;

      unsigned long long start_count = 0;           // an append starts behind the seeds already there
      if (append != NULL)
        start_count = (unsigned long long) (S->valid != NULL ? S->phys_count : S->count);

      hipEventRecord(dev->ev0,dev->stream);
      if (use_walk)
        { // every wavefront of the launch is resident (they are persistent): as many as the LDS of a CU holds, at most
          // the register budget's 4 x WAVE_OCC
          { const int lds = ((2*(XPC+1)*2 + (XT+4)*8 + XT*2 + XT + 16 + 1024 + A.wrawcap + 64 + 511) / 512) * 512;
            int per_cu = (160*1024) / lds;
            const int fit = per_cu;
            if (per_cu > 4*WAVE_OCC) per_cu = 4*WAVE_OCC;
            const char *ev = getenv("FGA_MERGE_WAVES");
            if (ev != NULL && atoi(ev) > 0 && atoi(ev) <= fit) per_cu = atoi(ev);
            if (grid > dev->ncu*per_cu) grid = dev->ncu*per_cu;
          }
          // ranges of equal merge cost, a few per wavefront, taken off a queue; cuts / queue head live in the tile area
          int nranges = grid*RANGES_PER_WAVE;
          if ((int64_t) nranges > total/(8*XT) + 1) nranges = (int) (total/(8*XT)) + 1;
          if (grid > nranges) grid = nranges;
          int64_t *cuts = (int64_t *) tiles;
          int *qhead = (int *) (wctr + 2);
          walk_args WA;
          WA.cuts = cuts; WA.nranges = nranges; WA.next = qhead;
          hipMemsetAsync(wctr,0,4*sizeof(unsigned long long),dev->stream);
          hipLaunchKernelGGL(range_cut_kernel,dim3((nranges+1+255)/256),dim3(256),0,dev->stream,
                             A.idx1,A.idx2,A.pbeg,A.pend,A.base,total,nranges,cuts);
          hipEventRecord(dev->ev1,dev->stream);
          if (self)
            hipLaunchKernelGGL(seed_merge_walk_kernel<MODE_SELF>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,WA);
          else if (prm->flip)
            hipLaunchKernelGGL(seed_merge_walk_kernel<MODE_FLIP>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,WA);
          else
            hipLaunchKernelGGL(seed_merge_walk_kernel<MODE_PAIR>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,WA);
        }
      else
      { int nb = (A.ntiles + 1 + 255) / 256;
        hipLaunchKernelGGL(merge_partition_kernel,dim3(nb),dim3(256),0,dev->stream,A,tiles);
        hipEventRecord(dev->ev1,dev->stream);
      }
      if (use_walk)
        ;
      else if (!use_wave)
        { if (self)
            hipLaunchKernelGGL(seed_merge_kernel<MODE_SELF>,dim3(grid),dim3(NT),0,dev->stream,A);
          else if (prm->flip)
            hipLaunchKernelGGL(seed_merge_kernel<MODE_FLIP>,dim3(grid),dim3(NT),0,dev->stream,A);
          else
            hipLaunchKernelGGL(seed_merge_kernel<MODE_PAIR>,dim3(grid),dim3(NT),0,dev->stream,A);
        }
      else
        { wave_out W;
          W.holes = holes; W.ctr = wctr; W.bigq = bigq; W.hole_cap = hole_cap; W.big_cap = big_cap;
          hipMemsetAsync(wctr,0,4*sizeof(unsigned long long),dev->stream);
          if (self)
            hipLaunchKernelGGL(seed_merge_wave_kernel<MODE_SELF>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,W);
          else if (prm->flip)
            hipLaunchKernelGGL(seed_merge_wave_kernel<MODE_FLIP>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,W);
          else
            hipLaunchKernelGGL(seed_merge_wave_kernel<MODE_PAIR>,dim3(grid),dim3(64),A.wrawcap,dev->stream,A,W);
          // the queued oversize tiles through the workgroup kernel (it reads their number from the device)
          hipLaunchKernelGGL(gather_big_tiles_kernel,dim3((big_cap+255)/256),dim3(256),0,dev->stream,
                             tiles,bigq,wctr+1,big_cap,pairs);
          merge_args B = A;
          B.tiles = pairs; B.pairs = 1; B.npairs = wctr+1; B.pair_cap = big_cap;
          int bgrid = dev->ncu * 4;
          if (bgrid > big_cap) bgrid = big_cap > 0 ? big_cap : 1;
          if (self)
            hipLaunchKernelGGL(seed_merge_kernel<MODE_SELF>,dim3(bgrid),dim3(NT),0,dev->stream,B);
          else if (prm->flip)
            hipLaunchKernelGGL(seed_merge_kernel<MODE_FLIP>,dim3(bgrid),dim3(NT),0,dev->stream,B);
          else
            hipLaunchKernelGGL(seed_merge_kernel<MODE_PAIR>,dim3(bgrid),dim3(NT),0,dev->stream,B);
        }

      // one round trip: counters, wave counters and the whole hole list land in pinned memory together
      const bool tm = getenv("FGA_MERGE_TIMING") != NULL;
      const double tm0 = tm ? fga_wall() : 0.;
      double tm1 = 0., tm2 = 0.;
      unsigned long long hw[4] = {0,0,0,0};
      unsigned long long *pin = (unsigned long long *) fga_dev_pinned(dev,sizeof(unsigned long long)*(8 + 2*(size_t) hole_cap)
                                                                          + sizeof(seed_move)*2*(size_t) (hole_cap+1));
      if (pin == NULL)
        { fga_set_error("fga_seed_merge: pinned staging allocation failed");
          goto done;
        }
      err = hipMemcpyAsync(pin,counters,2*sizeof(unsigned long long),hipMemcpyDeviceToHost,dev->stream);
      if (err == hipSuccess && use_walk)
        err = hipMemcpyAsync(pin+6,counters+2,sizeof(unsigned long long),hipMemcpyDeviceToHost,dev->stream);
      if (err == hipSuccess && use_wave && !use_walk)
        err = hipMemcpyAsync(pin+2,wctr,4*sizeof(unsigned long long),hipMemcpyDeviceToHost,dev->stream);
      if (err == hipSuccess && use_wave && !use_walk && hole_cap > 0)
        err = hipMemcpyAsync(pin+8,holes,sizeof(unsigned long long)*2*(size_t) hole_cap,hipMemcpyDeviceToHost,dev->stream);
      if (err == hipSuccess) err = hipStreamSynchronize(dev->stream);
      if (err == hipSuccess) err = hipGetLastError();
      if (err != hipSuccess)
        { fga_set_error("fga_seed_merge: kernel failed: %s",hipGetErrorString(err));
          goto done;
        }
      hc[0] = pin[0]; hc[1] = pin[1];
      if (use_wave && !use_walk) { hw[0] = pin[2]; hw[1] = pin[3]; }
      if (use_walk) hslots = (int64_t) pin[6];
      if (tm) tm1 = fga_wall();

      if (use_wave && (int64_t) hw[1] > big_cap)
        { // more oversize tiles than the queue holds (a pathologically repetitive input): redo everything with
          // the workgroup kernel
          hipMemcpyAsync(counters,&start_count,sizeof(unsigned long long),hipMemcpyHostToDevice,dev->stream);
          hipStreamSynchronize(dev->stream);
          fga_dev_release(dev,SLOT_TILES,work); work = NULL;
          use_wave = 0; use_walk = 0;
          continue;
        }

      if (use_wave && hw[0] > 0 && (int64_t) hc[0] <= phys)
        { // close the holes: seeds at the end of the allocated range move into the unused chunk tails
          const int nh = (int) (hw[0] < (unsigned long long) hole_cap ? hw[0] : (unsigned long long) hole_cap);
          const unsigned long long *hh = pin + 8;
          int64_t hsum = 0;
          for (int i = 0; i < nh; i++)
            hsum += (int64_t) (hh[2*(size_t) i+1] - hh[2*(size_t) i]);
          const int64_t C = (int64_t) hc[0], D = C - hsum;          // allocated, dense
          // destinations: the holes (clipped) below D, taken as they come; sources: the occupied stretches of [D,C),
          // which needs the few holes that reach above D in order.  Both lists are walked in step, no intermediate copies.
          static thread_local std::vector<std::pair<int64_t,int64_t> > above;
          above.clear();
          for (int i = 0; i < nh; i++)
            if ((int64_t) hh[2*(size_t) i+1] > D)
              above.push_back(std::make_pair((int64_t) hh[2*(size_t) i] > D ? (int64_t) hh[2*(size_t) i] : D,
                                             (int64_t) hh[2*(size_t) i+1]));
          std::sort(above.begin(),above.end());
          above.push_back(std::make_pair(C,C));                     // sentinel: the stretch after the last hole ends at C
          seed_move *mv = (seed_move *) (pin + 8 + 2*(size_t) hole_cap);
          size_t nmv = 0;
          { size_t ai = 0;
            int64_t spos = D, send = above[0].first;               // current source stretch [spos,send)
            for (int i = 0; i < nh; i++)
              { int64_t db = (int64_t) hh[2*(size_t) i], de = (int64_t) hh[2*(size_t) i+1];
                if (db >= D) continue;
                if (de > D) de = D;
                while (db < de)
                  { while (spos >= send && ai+1 < above.size())     // next occupied stretch above D
                      { spos = above[ai].second; ai += 1; send = above[ai].first; }
                    if (spos >= send)
                      break;
                    const int64_t l = (de-db) < (send-spos) ? (de-db) : (send-spos);
                    if (nmv >= 2*(size_t) (hole_cap+1))
                      { fga_set_error("fga_seed_merge: internal error, hole plan larger than its buffer");
                        goto done;
                      }
                    mv[nmv].src = spos; mv[nmv].dst = db; mv[nmv].len = l;
                    nmv += 1;
                    spos += l; db += l;
                  }
              }
          }
          if (nmv > 0)
            { if (hipMemcpyAsync(moves,mv,sizeof(seed_move)*nmv,hipMemcpyHostToDevice,dev->stream) != hipSuccess)
                { fga_set_error("fga_seed_merge: hole plan upload failed");
                  goto done;
                }
              hipLaunchKernelGGL(hole_fill_kernel,dim3((unsigned) nmv),dim3(256),0,dev->stream,S->seeds,moves,(int) nmv);
            }
          hc[0] = (unsigned long long) D;
          pin[0] = hc[0];
          hipMemcpyAsync(counters,pin,sizeof(unsigned long long),hipMemcpyHostToDevice,dev->stream);
        }
      hipEventRecord(ev2,dev->stream);
      if (tm) tm2 = fga_wall();
      if (hipStreamSynchronize(dev->stream) != hipSuccess)
        { fga_set_error("fga_seed_merge: hole fill failed: %s",hipGetErrorString(hipGetLastError()));
          goto done;
        }
      if (tm)
        fprintf(stderr,"merge host timing: launch->first sync %.1f us, plan+enqueue %.1f us, final sync %.1f us (%llu holes)\n",
                1e6*(tm1-tm0),1e6*(tm2-tm1),1e6*(fga_wall()-tm2),hw[0]);
      break;
    }
#ifdef MERGE_PROF
  { unsigned long long hp[8], z[8] = {0,0,0,0,0,0,0,0};
    hipMemcpyFromSymbol(hp,HIP_SYMBOL(merge_prof),sizeof(hp));
    hipMemcpyToSymbol(HIP_SYMBOL(merge_prof),z,sizeof(z));
    double tot = 0; for (int k = 0; k < 8; k++) tot += (double) hp[k];
    if (tot > 0)
      fprintf(stderr,"merge phases (%% of wave cycles): loop+idx wait %.1f  descriptor %.1f  issue %.1f  load wait %.1f  heads+keys %.1f  owner-scan %.1f  match %.1f  emit %.1f   (%.0f Mcycles)\n",
              100*hp[6]/tot,100*hp[7]/tot,100*hp[0]/tot,100*hp[1]/tot,100*hp[2]/tot,100*hp[3]/tot,100*hp[4]/tot,100*hp[5]/tot,tot*1e-6);
  }
#endif
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE_PARTITION],dev->ev0,dev->ev1);
  // the merge stage = everything the launch runs: range cuts / tile partition, the merge kernels, hole closing
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE],dev->ev0,ev2);
  S->phys_count = (int64_t) hc[0];
  S->count  = (int64_t) hc[0] - hslots;
  S->tseed  = (int64_t) hc[1];
  rc = 0;

done:
  if (ev2 != NULL) hipEventDestroy(ev2);
  fga_dev_release(dev,SLOT_TILES,work);
  if (rc != 0)
    { if (append == NULL)
        { hipFree(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); fga_dev_release(dev,SLOT_VALID,S->valid); free(S); }
      return 1;
    }
  if (out != NULL) *out = S;
  if (S->valid != NULL && S->phys_count > S->phys_capacity)
    S->count = S->phys_count;              // overflow of a block-allocated buffer: an upper bound of what is needed
  if (S->count > S->capacity || S->phys_count > S->phys_capacity)
    { fga_set_error("fga_seed_merge: %lld seeds exceed the buffer capacity %lld (re-run with a larger capacity)",
                    (long long) S->count,(long long) S->capacity);
      return 2;
    }
  return 0;
}

// cuts[0..nshards]: 12-mer prefix ranges [cuts[r], cuts[r+1]) of equal merge cost (entries of both tables + prefixes), the
// phase-1 shards of a multi-GPU run -- the reference splits its merge threads the same way (FastGA.c:2291-2321)
__global__ void prefix_cut_kernel(const int64_t *idx1, const int64_t *idx2, int64_t total, int nshards, int64_t *cuts)
{ const int w = blockIdx.x*blockDim.x + threadIdx.x;
  if (w > nshards)
    return;
  int64_t p = 0;
  if (w == nshards)
    p = FGA_NPREFIX;
  else if (w > 0)
    { const int64_t target = (total / nshards) * w;
      int lo = 0, hi = FGA_NPREFIX;
      while (lo < hi)                                  // smallest p whose inclusive cost exceeds the target
        { const int mid = lo + ((hi-lo) >> 1);
          const int64_t c = idx1[mid] + idx2[mid] + 2*((int64_t) mid+1);
          if (c > target) hi = mid; else lo = mid+1;
        }
      p = lo;
    }
  cuts[w] = p;
}

extern "C" int fga_merge_prefix_cuts(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2, int nshards, int64_t *cuts)
{ if (dev == NULL || t1 == NULL || cuts == NULL || nshards < 1 || nshards > 4096)
    { fga_set_error("fga_merge_prefix_cuts: bad argument");
      return 1;
    }
  if (t2 == NULL) t2 = t1;
  FGA_HIP(hipSetDevice(dev->device));
  int64_t c1e, c2e;
  FGA_HIP(hipMemcpy(&c1e,t1->index + (FGA_NPREFIX-1),8,hipMemcpyDeviceToHost));
  FGA_HIP(hipMemcpy(&c2e,t2->index + (FGA_NPREFIX-1),8,hipMemcpyDeviceToHost));
  const int64_t total = c1e + c2e + 2*(int64_t) FGA_NPREFIX;
  int64_t *d = (int64_t *) fga_dev_acquire(dev,SLOT_MISC,sizeof(int64_t)*(size_t) (nshards+1));
  if (d == NULL)
    { fga_set_error("fga_merge_prefix_cuts: device allocation failed");
      return 1;
    }
  hipLaunchKernelGGL(prefix_cut_kernel,dim3((nshards+1+63)/64),dim3(64),0,dev->stream,t1->index,t2->index,total,nshards,d);
  hipError_t e = hipMemcpyAsync(cuts,d,sizeof(int64_t)*(size_t) (nshards+1),hipMemcpyDeviceToHost,dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  fga_dev_release(dev,SLOT_MISC,d);
  if (e != hipSuccess)
    { fga_set_error("fga_merge_prefix_cuts: %s",hipGetErrorString(e));
      return 1;
    }
  for (int w = 1; w <= nshards; w++)
    if (cuts[w] < cuts[w-1]) cuts[w] = cuts[w-1];
  return 0;
}

extern "C" int fga_seed_merge(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                              const fga_merge_params *prm, int64_t capacity, fga_dseeds **out)
{ return merge_impl(dev,t1,t2,prm,capacity,out,NULL); }

extern "C" int fga_seed_merge_append(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                                     const fga_merge_params *prm, fga_dseeds *seeds)
{ if (seeds == NULL)
    { fga_set_error("fga_seed_merge_append: null seed buffer");
      return 1;
    }
  return merge_impl(dev,t1,t2,prm,0,NULL,seeds);
}
