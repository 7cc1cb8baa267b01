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
// LDS-staged tile (the common case), one 256-thread workgroup per tile:
//   1. the two index slices are loaded as 32-bit offsets relative to the tile start,
//   2. the raw on-disk bytes of both entry ranges stream HBM -> LDS with 16-byte-per-lane coalesced loads
//      (the tables are consumed in their on-disk 13..16-byte width, nothing is re-packed in HBM),
//   3. T2 entries are decoded once into 64-bit keys (suffix56 << 8 | mask) in LDS,
//   4. each lane takes T1 entries: panel lookup (binary search in the index slice), lower bound in the T2
//      panel, LCP with both neighbours by clz of the key xor, range growth bounded by FREQ,
//   5. seeds are appended with one global atomic per wavefront (wave-wide exclusive scan of lane counts).
// No MFMA anywhere: integer compare / byte work, HBM-bound by design.

#include "fga_device.hpp"

#define MERGE_THREADS   256
#define TILE_COST       1024                 // cost units per tile
#define PCAP            (TILE_COST/2 + 2)    // max prefixes of an LDS tile
#define KCAP            TILE_COST            // max T2 entries of an LDS tile
#define RAWCAP          (TILE_COST*16 + 64)  // bytes of raw entries staged per tile (E <= 16)

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
    const merge_tile *tiles;
    fga_seed *out; int64_t cap;
    unsigned long long *count;        // seeds produced
    unsigned long long *tseed;        // sum of plen (the reference's "ave. len" statistic)
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
    { int64_t target = A.base + (int64_t) w * TILE_COST;
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
__device__ __forceinline__ int lcp_suffix(uint64_t a, uint64_t b)      // a,b = 56-bit suffixes (bases 13..40)
{ uint64_t x = a ^ b;
  return x == 0 ? 40 : 12 + ((__clzll((long long) x) - 8) >> 1);
}

template <typename P>
__device__ __forceinline__ uint64_t load_be7(P e)                      // bytes 0..6 big-endian -> suffix56
{ return ((uint64_t) e[0] << 48) | ((uint64_t) e[1] << 40) | ((uint64_t) e[2] << 32)
       | ((uint64_t) e[3] << 24) | ((uint64_t) e[4] << 16) | ((uint64_t) e[5] << 8) | (uint64_t) e[6];
}

template <typename P>
__device__ __forceinline__ uint32_t load_le(P e, int n)
{ uint32_t v = 0;
  for (int k = 0; k < n; k++)
    v |= (uint32_t) e[k] << (8*k);
  return v;
}

// wave-wide exclusive scan of a small non-negative count; returns the lane's offset, total in `total`
__device__ __forceinline__ int wave_excl_scan(int v, int &total)
{ int lane = threadIdx.x & 63;
  int x = v;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { int y = __shfl_up(x,d,64);
      if (lane >= d) x += y;
    }
  total = __shfl(x,63,64);
  return x - v;
}

// Everything needed about one T1 entry and its matching T2 run
struct hit_t
  { int plen;
    int cnt;        // seeds this entry emits
    int64_t low, hgh;
  };

// ---------------------------------------------------------------------------------------------------
// Accessors: the same matching code runs over an LDS-staged tile or straight from global memory
// ---------------------------------------------------------------------------------------------------
struct lds_t2
  { const uint64_t *key;     // decoded suffix56<<8 | mask
    const uint8_t  *raw;     // raw entries, entry j at raw + j*E
    int E, post, cont;
    __device__ __forceinline__ uint64_t suffix(int64_t j) const { return key[j] >> 8; }
    __device__ __forceinline__ int      mask(int64_t j)   const { return (int) (key[j] & 0xff); }
    __device__ __forceinline__ void payload(int64_t j, uint32_t &pos, uint32_t &ctg, uint32_t &sign) const
    { const uint8_t *e = raw + j*E + 9;
      pos = load_le(e,post);
      uint32_t c = load_le(e+post,cont);
      uint32_t sb = 0x80u << (8*(cont-1));
      sign = (c & sb) != 0;
      ctg  = c & (sb-1);
    }
  };

struct glb_t2
  { const uint8_t *tab;      // entry j (absolute index) at tab + j*E
    int E, post, cont;
    __device__ __forceinline__ uint64_t suffix(int64_t j) const { return load_be7(tab + j*E); }
    __device__ __forceinline__ int      mask(int64_t j)   const { return tab[j*E+7]; }
    __device__ __forceinline__ void payload(int64_t j, uint32_t &pos, uint32_t &ctg, uint32_t &sign) const
    { const uint8_t *e = tab + j*E + 9;
      pos = load_le(e,post);
      uint32_t c = load_le(e+post,cont);
      uint32_t sb = 0x80u << (8*(cont-1));
      sign = (c & sb) != 0;
      ctg  = c & (sb-1);
    }
  };

// pair mode: match T1 suffix ks against T2 panel [b0,b1)
template <typename T2>
__device__ __forceinline__ void match_pair(const T2 &t2, uint64_t ks, int64_t b0, int64_t b1, int freq, hit_t &h)
{ int64_t lo = b0, hi = b1;
  while (lo < hi)
    { int64_t m = (lo+hi) >> 1;
      if (t2.suffix(m) < ks) lo = m+1; else hi = m;
    }
  int la = (lo > b0) ? lcp_suffix(ks,t2.suffix(lo-1)) : 0;
  int lc = (lo < b1) ? lcp_suffix(ks,t2.suffix(lo)) : 0;
  int plen = la > lc ? la : lc;
  int64_t low = lo, hgh = lo;
  while (low > b0 && lo-low <= freq && lcp_suffix(ks,t2.suffix(low-1)) >= plen)
    low -= 1;
  while (hgh < b1 && hgh-low <= freq && lcp_suffix(ks,t2.suffix(hgh)) >= plen)
    hgh += 1;
  h.plen = plen; h.low = low; h.hgh = hgh;
}

// self mode: entry k of panel [a0,a1); plen = max(lcp with predecessor, lcp with successor)
template <typename T2>
__device__ __forceinline__ void match_self(const T2 &t2, int64_t k, int64_t a0, int64_t a1, int freq, hit_t &h)
{ uint64_t ks = t2.suffix(k);
  int lk  = (k > a0)   ? lcp_suffix(ks,t2.suffix(k-1)) : 0;
  int lk1 = (k+1 < a1) ? lcp_suffix(ks,t2.suffix(k+1)) : 11;
  int plen = lk > lk1 ? lk : lk1;
  int64_t low = k, hgh = k+1;
  while (low > a0 && k-low <= freq && lcp_suffix(ks,t2.suffix(low-1)) >= plen)
    low -= 1;
  while (hgh < a1 && hgh-low <= freq && lcp_suffix(ks,t2.suffix(hgh)) >= plen)
    hgh += 1;
  h.plen = plen; h.low = low; h.hgh = hgh;
}

// Emit (or count, when out == nullptr-equivalent pass) the seeds of one T1 entry.
//   s*  : T1 entry fields;  selfk : index of the entry itself inside t2 (self mode) or -1
template <typename T2, bool COUNT>
__device__ __forceinline__ int emit_entry(const merge_args &A, const T2 &t2, const hit_t &h,
                                          int smask, uint32_t spos, uint32_t sctg, uint32_t ssign,
                                          int64_t selfk, fga_seed *out, int64_t cap, int64_t wpos)
{ if (h.hgh - h.low >= A.freq)
    return 0;
  int mlen = A.soft_mask ? h.plen : 41;
  if (smask >= mlen)
    return 0;
  if (!A.self && !A.flip && ssign)
    return 0;
  int n = 0;
  for (int64_t j = h.low; j < h.hgh; j++)
    { if (j == selfk || t2.mask(j) >= mlen)
        continue;
      uint32_t cpos, cctg, csign;
      t2.payload(j,cpos,cctg,csign);
      if (A.flip && csign)
        continue;
      if (!COUNT)
        { fga_seed sd;
          if (A.flip)            // table 1 is genome 2: A side = c (forward), B side = s
            { sd.apos = cpos; sd.bpos = spos;
              sd.actg = (cctg << 8) | (uint32_t) h.plen;
              sd.bctg = sctg | (ssign << 30) | (ssign << 31);
            }
          else if (A.self)       // stream N iff the signs agree; A payload goes out with its sign cleared
            { sd.apos = spos; sd.bpos = cpos;
              sd.actg = (sctg << 8) | (uint32_t) h.plen;
              sd.bctg = cctg | (csign << 30) | ((uint32_t) (ssign != csign) << 31);
            }
          else
            { sd.apos = spos; sd.bpos = cpos;
              sd.actg = (sctg << 8) | (uint32_t) h.plen;
              sd.bctg = cctg | (csign << 30) | (csign << 31);
            }
          if (wpos+n < cap)
            out[wpos+n] = sd;
        }
      n += 1;
    }
  return n;
}

// Wave-aggregated append: every lane of the wave calls this with its own entry (or cnt = 0).
template <typename T2>
__device__ __forceinline__ void wave_append(const merge_args &A, const T2 &t2, const hit_t &h, bool valid,
                                            int smask, uint32_t spos, uint32_t sctg, uint32_t ssign, int64_t selfk)
{ int cnt = valid ? emit_entry<T2,true>(A,t2,h,smask,spos,sctg,ssign,selfk,nullptr,0,0) : 0;
  int total;
  int off = wave_excl_scan(cnt,total);
  if (total == 0)
    return;
  unsigned long long base = 0;
  int lane = threadIdx.x & 63;
  int tl = cnt * h.plen, tsum = tl;
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0)
    { base = atomicAdd(A.count,(unsigned long long) total);
      atomicAdd(A.tseed,(unsigned long long) tsum);
    }
  base = __shfl(base,0,64);
  if (cnt > 0)
    emit_entry<T2,false>(A,t2,h,smask,spos,sctg,ssign,selfk,A.out,A.cap,(int64_t) base + off);
}

// ---------------------------------------------------------------------------------------------------
// the merge kernel
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MERGE_THREADS)
void seed_merge_kernel(merge_args A)
{ __shared__ uint32_t la[PCAP+1];            // la[q] = #T1 entries of the tile in prefixes <= p0+q
  __shared__ uint32_t lb[PCAP+1];
  __shared__ __attribute__((aligned(16))) uint8_t  raw[RAWCAP];
  __shared__ __attribute__((aligned(16))) uint64_t keyB[KCAP];

  const int tid = threadIdx.x;
  const merge_tile t0 = A.tiles[blockIdx.x];
  const merge_tile t1 = A.tiles[blockIdx.x+1];
  const int     p0 = t0.p, p1 = t1.p;
  const int     np = p1 - p0;
  if (np <= 0)
    return;
  const int64_t a0 = t0.a, a1 = t1.a;
  const int64_t b0 = A.self ? a0 : t0.b, b1 = A.self ? a1 : t1.b;
  const int64_t n1 = a1 - a0, n2 = b1 - b0;
  if (n1 == 0 || n2 == 0)
    return;

  const int E1 = A.E1, E2 = A.E2;

  // byte extents, aligned down to 16 for the coalesced copy
  const int64_t s1 = a0*E1, e1 = a1*E1;
  const int64_t s2 = b0*E2, e2 = b1*E2;
  const int64_t s1a = s1 & ~(int64_t) 15, s2a = s2 & ~(int64_t) 15;
  const int64_t len1 = ((e1 - s1a) + 15) & ~(int64_t) 15;
  const int64_t len2 = A.self ? 0 : (((e2 - s2a) + 15) & ~(int64_t) 15);

  const bool fits = (np <= PCAP) && (n2 <= KCAP) && (len1 + len2 <= RAWCAP);

  if (fits)
    { // 1. index slices
      for (int q = tid; q < np; q += MERGE_THREADS)
        { la[q] = (uint32_t) (A.idx1[p0+q] - a0);
          lb[q] = A.self ? la[q] : (uint32_t) (A.idx2[p0+q] - b0);
        }
      // 2. raw bytes, 16 B per lane
      { const uint4 *g1 = (const uint4 *) (A.tab1 + s1a);
        uint4 *l1 = (uint4 *) raw;
        int n16 = (int) (len1 >> 4);
        for (int x = tid; x < n16; x += MERGE_THREADS)
          l1[x] = g1[x];
        if (!A.self)
          { const uint4 *g2 = (const uint4 *) (A.tab2 + s2a);
            uint4 *l2 = (uint4 *) (raw + len1);
            int m16 = (int) (len2 >> 4);
            for (int x = tid; x < m16; x += MERGE_THREADS)
              l2[x] = g2[x];
          }
      }
      __syncthreads();

      const uint8_t *r1 = raw + (s1 - s1a);
      const uint8_t *r2 = A.self ? r1 : raw + len1 + (s2 - s2a);

      // 3. T2 keys
      for (int j = tid; j < (int) n2; j += MERGE_THREADS)
        { const uint8_t *e = r2 + j*E2;
          keyB[j] = (load_be7(e) << 8) | e[7];
        }
      __syncthreads();

      lds_t2 t2;
      t2.key = keyB; t2.raw = r2; t2.E = E2; t2.post = A.post2; t2.cont = A.cont2;

      // 4./5. one T1 entry per lane per round; whole waves stay in the loop for the wave-wide append
      const int rounds = ((int) n1 + MERGE_THREADS - 1) / MERGE_THREADS;
      for (int r = 0; r < rounds; r++)
        { int  i = r*MERGE_THREADS + tid;
          bool valid = i < (int) n1;
          hit_t h; h.plen = 0; h.cnt = 0; h.low = h.hgh = 0;
          int smask = 0; uint32_t spos = 0, sctg = 0, ssign = 0;
          int64_t selfk = -1;
          if (valid)
            { // panel of entry i: smallest q with la[q] > i
              int lo = 0, hi = np-1;
              while (lo < hi)
                { int m = (lo+hi) >> 1;
                  if (la[m] > (uint32_t) i) hi = m; else lo = m+1;
                }
              int q = lo;
              int64_t pb0 = q ? lb[q-1] : 0, pb1 = lb[q];
              if (pb0 == pb1)
                valid = false;
              else
                { const uint8_t *e = r1 + i*E1;
                  smask = e[7];
                  spos  = load_le(e+9,A.post1);
                  uint32_t c  = load_le(e+9+A.post1,A.cont1);
                  uint32_t sb = 0x80u << (8*(A.cont1-1));
                  ssign = (c & sb) != 0;
                  sctg  = c & (sb-1);
                  if (A.self)
                    { selfk = i;
                      match_self(t2,(int64_t) i,pb0,pb1,A.freq,h);
                    }
                  else
                    match_pair(t2,load_be7(e),pb0,pb1,A.freq,h);
                }
            }
          wave_append(A,t2,h,valid,smask,spos,sctg,ssign,selfk);
        }
    }
  else
    { // Oversize tile (one huge panel, possibly with a few neighbours): same logic straight from HBM.
      glb_t2 t2;
      t2.tab = A.self ? A.tab1 : A.tab2; t2.E = E2; t2.post = A.post2; t2.cont = A.cont2;
      const int64_t rounds = (n1 + MERGE_THREADS - 1) / MERGE_THREADS;
      for (int64_t r = 0; r < rounds; r++)
        { int64_t i = a0 + r*MERGE_THREADS + tid;      // absolute T1 entry
          bool valid = i < a1;
          hit_t h; h.plen = 0; h.cnt = 0; h.low = h.hgh = 0;
          int smask = 0; uint32_t spos = 0, sctg = 0, ssign = 0;
          int64_t selfk = -1;
          if (valid)
            { int lo = p0, hi = p1-1;                  // smallest p with idx1[p] > i
              while (lo < hi)
                { int m = lo + ((hi-lo) >> 1);
                  if (A.idx1[m] > i) hi = m; else lo = m+1;
                }
              int64_t pb0, pb1;
              if (A.self)
                { pb0 = idx_at(A.idx1,lo-1); pb1 = A.idx1[lo]; }
              else
                { pb0 = idx_at(A.idx2,lo-1); pb1 = A.idx2[lo]; }
              if (pb0 == pb1)
                valid = false;
              else
                { const uint8_t *e = A.tab1 + i*E1;
                  smask = e[7];
                  spos  = load_le(e+9,A.post1);
                  uint32_t c  = load_le(e+9+A.post1,A.cont1);
                  uint32_t sb = 0x80u << (8*(A.cont1-1));
                  ssign = (c & sb) != 0;
                  sctg  = c & (sb-1);
                  if (A.self)
                    { selfk = i;
                      match_self(t2,i,pb0,pb1,A.freq,h);
                    }
                  else
                    match_pair(t2,load_be7(e),pb0,pb1,A.freq,h);
                }
            }
          wave_append(A,t2,h,valid,smask,spos,sctg,ssign,selfk);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
extern "C" int fga_seed_merge(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                              const fga_merge_params *prm, int64_t capacity, fga_dseeds **out)
{ *out = NULL;
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
  A.ntiles = (int) (total / TILE_COST) + 1;

  fga_dseeds *S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
  if (S == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  S->dev = dev;
  if (capacity <= 0)
    capacity = 2*(c1e - c1b) + (1<<20);
  S->capacity = capacity;

  merge_tile *tiles = NULL;
  unsigned long long *counters = NULL;
  hipError_t err;
  if ((err = hipMalloc(&tiles,sizeof(merge_tile)*(size_t) (A.ntiles+1))) != hipSuccess ||
      (err = hipMalloc(&counters,2*sizeof(unsigned long long))) != hipSuccess ||
      (err = hipMalloc(&S->seeds,sizeof(fga_seed)*(size_t) capacity)) != hipSuccess)
    { fga_set_error("fga_seed_merge: device allocation failed: %s",hipGetErrorString(err));
      hipFree(tiles); hipFree(counters); hipFree(S->seeds); free(S);
      return 1;
    }
  S->dcount = (int64_t *) counters;
  A.tiles = tiles; A.out = S->seeds; A.cap = capacity;
  A.count = counters; A.tseed = counters+1;

  hipMemsetAsync(counters,0,2*sizeof(unsigned long long),dev->stream);
  hipEventRecord(dev->ev0,dev->stream);
  { int nb = (A.ntiles + 1 + 255) / 256;
    hipLaunchKernelGGL(merge_partition_kernel,dim3(nb),dim3(256),0,dev->stream,A,tiles);
  }
  hipEventRecord(dev->ev1,dev->stream);
  hipLaunchKernelGGL(seed_merge_kernel,dim3(A.ntiles),dim3(MERGE_THREADS),0,dev->stream,A);
  hipEvent_t ev2;
  hipEventCreate(&ev2);
  hipEventRecord(ev2,dev->stream);
  unsigned long long hc[2];
  err = hipMemcpyAsync(hc,counters,sizeof(hc),hipMemcpyDeviceToHost,dev->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(dev->stream);
  if (err == hipSuccess) err = hipGetLastError();
  if (err != hipSuccess)
    { fga_set_error("fga_seed_merge: kernel failed: %s",hipGetErrorString(err));
      hipEventDestroy(ev2);
      hipFree(tiles); hipFree(counters); hipFree(S->seeds); free(S);
      return 1;
    }
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE_PARTITION],dev->ev0,dev->ev1);
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE],dev->ev1,ev2);
  hipEventDestroy(ev2);
  hipFree(tiles);
  S->count  = (int64_t) hc[0];
  S->tseed  = (int64_t) hc[1];
  *out = S;
  if (S->count > S->capacity)
    { fga_set_error("fga_seed_merge: %lld seeds exceed the buffer capacity %lld (re-run with a larger capacity)",
                    (long long) S->count,(long long) S->capacity);
      return 2;
    }
  return 0;
}
