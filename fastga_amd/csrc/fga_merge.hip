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
// The tables are read through their field-per-array views (fga_view.hip): 64-bit keys that carry the low byte of the
// 12-mer prefix above the 56-bit suffix, lcp / mask bytes, positions and contig|sign words as arrays of their own.
// One launch does the whole merge:
//   * range_cut_kernel cuts the prefix space into ranges of equal cost, a few per wavefront; wavefronts (one per
//     workgroup, persistent) take ranges off a queue and WALK them: the prefix-index entries of the next 64 prefixes sit
//     one per lane (fetched a tile ahead, HBM -> LDS directly), a ballot finds how many whole panels fit a tile -- at
//     most T1CAP entries of table 1 and T2CAP of table 2, never across a multiple of 256 prefixes, so that the keys of a
//     tile order like its 40-mers.
//   * a tile: T2 keys / lcp bytes / payloads and the T1 payloads travel HBM -> LDS with global_load_lds (no register
//     staging, no unpacking); the T1 keys go to registers, one entry per lane and round.  Every T1 entry finds its lower
//     bound among ALL T2 keys of the tile (panels need no bookkeeping: a key of another panel differs in its top byte, and
//     the lcp byte of a panel's first entry is < 12 <= plen, so neither the neighbour test nor the run growth ever leaves
//     the panel); the searches of the up to four rounds run side by side (independent LDS chains in flight together);
//     LCP with both neighbours is a clz of the key xor; the run grows on the table's own lcp bytes, bounded by FREQ.
//   * emission is seed-parallel: descriptors at the first slot of each run, a wave max-scan, 16-byte stores into
//     1024-slot blocks taken with one atomic per block; unused block tails stay open (valid[] per block).
//   * a panel that does not fit a tile (repeat families; every panel of a 3 Gbp table) is STREAMED: windows of the T2
//     panel with the T1 entries whose lower bound lies at least FREQ+2 entries inside the window (as far as the run growth
//     ever looks); the window then moves to the last consumed entry's lower bound.  No search outside LDS unless a
//     window holds no T1 key at all.
// In a pair comparison table 1 is read through its FORWARD view: its complement-strand entries emit nothing
// (FastGA.c:921-928) and are not read.  No MFMA anywhere: integer compare / byte work, HBM-bound by design.

#include "fga_device.hpp"

enum { MODE_PAIR = 0, MODE_FLIP = 1, MODE_SELF = 2 };

#define T1CAP           256                  // table-1 entries per tile: four rounds of one entry per lane
#define XPC             64                   // prefixes per tile at most (one index entry per lane)
#define EWIN            512                  // slots per emission window (descriptor dwords in the key array)
#ifndef WAVE_OCC
#define WAVE_OCC        5                    // resident wavefronts per SIMD the register budget is held to
#endif
#ifndef RANGES_PER_WAVE
#define RANGES_PER_WAVE 4
#endif

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL,"wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define VM_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70)      // vmcnt(0), the other counters left alone

struct merge_args
  { fga_view v1, v2;                  // table 1 (pair mode: its forward view) and table 2 (self: the same table)
    uint32_t sign1, sign2;            // sign bit of the contig words
    int   freq, soft_mask;
    int   pbeg, pend;                 // prefix range handled by this call
    int64_t base;                     // cost(pbeg-1)
    fga_seed *out; int64_t cap;
    unsigned long long *count;        // slots handed out
    unsigned long long *tseed;        // sum of plen (the reference's "ave. len" statistic)
    unsigned long long *hslots;       // slots left unused
    uint16_t *valid;                  // seeds per 1024-slot block of `out` (holes are left open, consumers skip them)
    int64_t   nblocks;
    const int64_t *cuts;              // [nranges+1] prefix boundaries
    int       nranges;
    int      *next;                   // range queue head
  };

__global__ void range_cut_kernel(const uint32_t *idx1, const uint32_t *idx2, int pbeg, int pend, int64_t base,
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
          const int64_t c = (int64_t) idx1[mid] + (int64_t) idx2[mid] + 2*((int64_t) mid+1);
          if (c > target) hi = mid; else lo = mid+1;
        }
      p = lo;
    }
  cuts[w] = p;
}

// LCP in bases of two keys (prefix byte | 56-bit suffix): 0 when the prefixes differ, else 12 .. 40
__device__ __forceinline__ int lcp_key(uint64_t a, uint64_t b)
{ const uint64_t x = a ^ b;
  if (x >> 56) return 0;
  return x == 0 ? 40 : 8 + (__clzll((long long) x) >> 1);       // clz >= 8; 12 + (clz - 8)/2
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

// smallest j in [lo,hi] with K[j] >= kq, all 64 lanes probing HBM: the range shrinks 64-fold a round (only when a window
// of an oversize panel holds no T1 key)
__device__ __forceinline__ int64_t wave_lower_bound(const uint64_t *K, int64_t lo, int64_t hi, uint64_t kq)
{ const int lane = threadIdx.x;
  while (hi > lo)
    { const int64_t step = ((hi - lo) + 63) >> 6;
      const int64_t pos = lo + (int64_t) lane*step;
      bool below = false;
      if (pos < hi)
        below = K[pos] < kq;
      const int t = __popcll(__builtin_amdgcn_ballot_w64(below));        // sorted: the first t probes are below
      if (t == 0)
        return lo;
      const int64_t nhi = lo + (int64_t) t*step;
      lo = lo + (int64_t) (t-1)*step + 1;
      if (nhi < hi) hi = nhi;
    }
  return lo;
}

struct walk_out                      // a wavefront's current output chunk and statistics (wave-uniform)
  { int64_t chunk_pos, chunk_end;
    unsigned long long tsum;
  };

// the LDS of one wavefront
template <int T2CAP>
struct tile_lds
  { uint64_t keyB0[T2CAP + 8];       // keyB = keyB0 + 4 + (window start & 3): T2 keys, one spare either side; emission descriptors later
    uint32_t pB0[T2CAP + 8];         // T2 positions
    uint32_t pA0[T1CAP + 8];         // T1 positions
    uint8_t  lcpB0[T2CAP + 24];      // T2 lcp bytes (8 spare bytes in front)
    uint8_t  mB0[T2CAP + 8];         // T2 mask bytes  (soft mask runs only)
    uint8_t  mA0[T1CAP + 8];         // T1 mask bytes
    uint32_t ixs[2][64];             // index entries of the next 64 prefixes
  };

__device__ __forceinline__ uint32_t lds_c(const uint8_t *c, int cw, int i)
{ if (cw == 1) return c[i];
  if (cw == 2) return ((const uint16_t *) c)[i];
  return ((const uint32_t *) c)[i];
}

#define G2L(gp,lp,sz) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (gp), \
                                                       (__attribute__((address_space(3))) void *) (lp),sz,0,0)

// One tile: T1 entries [a0, a0+n1) against the T2 window [b0, b0+n2) (self: one window, its entries [t_lo, t_hi) emit).
// limit: the window does not reach the end of its panel, so only the T1 entries whose lower bound is at most n2 - margin
// are consumed (a prefix of them).  Returns their number; lb_last = lower bound (window coordinates) of the last one.
template <int MODE, int T2CAP, int NR>
__device__ __forceinline__ void match_rounds(const merge_args &A, const uint64_t *keyB, const uint8_t *lcpB, const uint8_t *mA,
                                             const uint8_t *mB, const uint8_t *cB, const uint64_t *k1, int n2, int na, int t_lo,
                                             uint32_t *res, int &total, unsigned long long &tsum, int &lb_last)
{ const int lane = threadIdx.x;
  const int freq = A.freq;
  uint64_t ks[NR];
  int base[NR];
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { ks[r] = (MODE == MODE_SELF) ? keyB[r*64 + lane < na ? t_lo + r*64 + lane : t_lo] : k1[r];
      base[r] = 0;
    }
  if (MODE != MODE_SELF)
    { // lower bounds over the whole window: a wave-uniform number of halving steps, the chains of all rounds in flight together
      int len = n2;
      while (len > 1)
        { const int half = len >> 1;
          #pragma unroll
          for (int r = 0; r < NR; r++)
            base[r] += keyB[base[r] + half - 1] < ks[r] ? half : 0;
          len -= half;
        }
      #pragma unroll
      for (int r = 0; r < NR; r++)
        base[r] += keyB[base[r]] < ks[r] ? 1 : 0;
    }
  uint64_t kb[NR], kc[NR];
  int bd[NR], bu[NR], nb[NR], na_[NR];
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { if (MODE == MODE_SELF) { const int i = r*64 + lane < na ? t_lo + r*64 + lane : t_lo; nb[r] = i-1; na_[r] = i+1; base[r] = i; }
      else                   { nb[r] = base[r]-1; na_[r] = base[r]; }
      kb[r] = keyB[nb[r]]; kc[r] = keyB[na_[r]];                  // keyB[-1] and keyB[n2] hold keys of no panel
      bd[r] = (int) lcpB[nb[r]]; bu[r] = (int) lcpB[na_[r]+1];
    }
  total = 0;
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { const int c = r*64 + lane;
      const bool act = c < na;
      const int i = (MODE == MODE_SELF) ? (act ? t_lo + c : t_lo) : c;
      int low, hgh, lbnd;
      if (MODE == MODE_SELF) { low = i; hgh = i+1; lbnd = i; }
      else                   { low = hgh = lbnd = base[r]; }
      const int lkb = lcp_key(ks[r],kb[r]), lka = lcp_key(ks[r],kc[r]);
      const int plen = lkb > lka ? lkb : lka;                      // 0: no T2 entry of this panel next to the key
      const bool ok = act && plen >= 12;
      // run growth on the table's own lcp bytes; the first step of either direction is decided on the bytes read above
      const bool gd = ok && lkb >= plen;
      low -= gd ? 1 : 0;
      if (gd && low > 0 && lbnd-low <= freq && bd[r] >= plen)
        { low -= 1;
          while (low > 0 && lbnd-low <= freq && (int) lcpB[low] >= plen)
            low -= 1;
        }
      const bool gu = ok && lka >= plen && hgh < n2 && hgh-low <= freq;
      hgh += gu ? 1 : 0;
      if (gu && hgh < n2 && hgh-low <= freq && bu[r] >= plen)
        { hgh += 1;
          while (hgh < n2 && hgh-low <= freq && (int) lcpB[hgh] >= plen)
            hgh += 1;
        }
      const int mlen = A.soft_mask ? plen : 41;
      bool pass = ok && hgh-low < freq;
      if (A.soft_mask)
        pass = pass && (int) (MODE == MODE_SELF ? mB[i] : mA[i]) < mlen;
      int cnt;
      if (MODE == MODE_FLIP || A.soft_mask)
        { cnt = 0;
          if (pass)
            for (int j = low; j < hgh; j++)
              { if (A.soft_mask && (int) mB[j] >= mlen)
                  continue;
                if (MODE == MODE_FLIP && (lds_c(cB,A.v2.cw,j) & A.sign2))
                  continue;
                if (MODE == MODE_SELF && j == i)
                  continue;
                cnt += 1;
              }
        }
      else
        cnt = pass ? (hgh-low) - (MODE == MODE_SELF ? 1 : 0) : 0;
      res[r] = (pass && cnt > 0) ? ((uint32_t) i | ((uint32_t) low << 8) | ((uint32_t) plen << 18) | ((uint32_t) cnt << 24)) : 0u;
      total += cnt;
      tsum += (unsigned long long) cnt * plen;
    }
  // lower bound of the last consumed entry (entry na-1: round (na-1) >> 6, lane (na-1) & 63)
  lb_last = 0;
  if (MODE != MODE_SELF && na > 0)
    { const int lr = (na-1) >> 6, ll = (na-1) & 63;
      #pragma unroll
      for (int r = 0; r < NR; r++)
        if (r == lr)
          lb_last = __builtin_amdgcn_readlane(base[r],ll);
    }
  #pragma unroll
  for (int r = NR; r < 4; r++)
    res[r] = 0;
}

template <int MODE, int T2CAP>
__device__ __forceinline__ int walk_tile(const merge_args &A, tile_lds<T2CAP> &S, uint8_t *cdyn, int64_t a0, int n1,
                                         int64_t b0, int n2, int t_lo, int t_hi, bool limit, int margin, walk_out &O,
                                         int &lb_last)
{ const int lane = threadIdx.x;
  const fga_view &V1 = A.v1, &V2 = A.v2;
  const int cw1 = V1.cw, cw2 = V2.cw;
  const int ob = (int) (b0 & 3), oa = (int) (a0 & 3);
  const int64_t b0a = b0 - ob, a0a = a0 - oa;
  uint64_t *keyB = S.keyB0 + 4 + ob;
  uint32_t *pB = S.pB0 + ob, *pA = S.pA0 + oa;
  uint8_t  *lcpB = S.lcpB0 + 8 + ob, *mB = S.mB0 + ob, *mA = S.mA0 + oa;
  uint8_t  *cB0 = cdyn, *cA0 = cdyn + (size_t) (T2CAP + 8)*cw2;
  uint8_t  *cB = cB0 + (size_t) ob*cw2, *cA = cA0 + (size_t) oa*cw1;
  uint32_t *own32 = (uint32_t *) S.keyB0;          // the emission reuses the key array (keys are done with by then)

  // 1. HBM -> LDS (global_load_lds): T2 keys, lcp bytes, payloads; T1 payloads.  T1 keys -> registers.
  { const int m2 = n2 + ob;                          // staged T2 entries, from the 4-entry aligned start
    const uint4 *gk = (const uint4 *) (V2.K + b0a);
    uint4 *lk = (uint4 *) (S.keyB0 + 4);
    for (int x = lane; 2*x < m2; x += 64)
      G2L(gk + x,lk + x,16);
    const uint4 *gp = (const uint4 *) (V2.P + b0a);
    uint4 *lp = (uint4 *) S.pB0;
    for (int x = lane; 4*x < m2; x += 64)
      G2L(gp + x,lp + x,16);
    const uint32_t *gl = (const uint32_t *) (V2.L + b0a);
    uint32_t *ll = (uint32_t *) (S.lcpB0 + 8);
    for (int x = lane; 4*x < m2 + 2; x += 64)        // two lcp bytes beyond the window are read (never used)
      G2L(gl + x,ll + x,4);
    const uint32_t *gc = (const uint32_t *) ((const uint8_t *) V2.C + (size_t) b0a*cw2);
    uint32_t *lc = (uint32_t *) cB0;
    for (int x = lane; 4*x < m2*cw2; x += 64)
      G2L(gc + x,lc + x,4);
    if (A.soft_mask)
      { const uint32_t *gm = (const uint32_t *) (V2.M + b0a);
        uint32_t *lm = (uint32_t *) S.mB0;
        for (int x = lane; 4*x < m2; x += 64)
          G2L(gm + x,lm + x,4);
      }
    if (MODE != MODE_SELF)
      { const int m1 = n1 + oa;
        const uint4 *gpa = (const uint4 *) (V1.P + a0a);
        uint4 *lpa = (uint4 *) S.pA0;
        for (int x = lane; 4*x < m1; x += 64)
          G2L(gpa + x,lpa + x,16);
        const uint32_t *gca = (const uint32_t *) ((const uint8_t *) V1.C + (size_t) a0a*cw1);
        uint32_t *lca = (uint32_t *) cA0;
        for (int x = lane; 4*x < m1*cw1; x += 64)
          G2L(gca + x,lca + x,4);
        if (A.soft_mask)
          { const uint32_t *gma = (const uint32_t *) (V1.M + a0a);
            uint32_t *lma = (uint32_t *) S.mA0;
            for (int x = lane; 4*x < m1; x += 64)
              G2L(gma + x,lma + x,4);
          }
      }
  }
  uint64_t k1[4];
  #pragma unroll
  for (int r = 0; r < 4; r++)
    { const int i = r*64 + lane;
      k1[r] = (MODE != MODE_SELF) ? V1.K[a0 + (i < n1 ? i : n1-1)] : 0;
    }
  // the direct-to-LDS loads are asynchronous and nothing below depends on a VGPR they return: wait for them by hand
  VM_WAIT();
  WSYNC();
  // keys of no panel either side of the window (the real neighbours there may share the low prefix byte)
  if (lane == 0)
    { const uint64_t sent = keyB[0] ^ 0x8000000000000000ull;
      keyB[-1] = sent; keyB[n2] = sent;
      lcpB[n2] = 0; lcpB[n2+1] = 0;
    }
  WSYNC();

  // 2. which T1 entries this window can finish
  int na = (MODE == MODE_SELF) ? t_hi - t_lo : n1;
  if (MODE != MODE_SELF && limit)
    { const uint64_t kl = keyB[n2 - margin];       // lower bound <= n2 - margin  <=>  key <= this one
      na = 0;
      #pragma unroll
      for (int r = 0; r < 4; r++)
        na += __popcll(__builtin_amdgcn_ballot_w64(r*64 + lane < n1 && k1[r] <= kl));
    }
  lb_last = 0;
  if (na <= 0)
    { WSYNC();
      return 0;
    }

  // 3. match: result per round packed i (8 bits) | low << 8 | plen << 18 | seeds << 24
  uint32_t res[4];
  int total = 0;
  { const int nr = (na + 63) >> 6;
    if (nr == 1)      match_rounds<MODE,T2CAP,1>(A,keyB,lcpB,mA,mB,cB,k1,n2,na,t_lo,res,total,O.tsum,lb_last);
    else if (nr == 2) match_rounds<MODE,T2CAP,2>(A,keyB,lcpB,mA,mB,cB,k1,n2,na,t_lo,res,total,O.tsum,lb_last);
    else if (nr == 3) match_rounds<MODE,T2CAP,3>(A,keyB,lcpB,mA,mB,cB,k1,n2,na,t_lo,res,total,O.tsum,lb_last);
    else              match_rounds<MODE,T2CAP,4>(A,keyB,lcpB,mA,mB,cB,k1,n2,na,t_lo,res,total,O.tsum,lb_last);
  }

  // 4. slots and emission
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
      // Seed-parallel, in windows of EWIN slots: every entry with seeds in the window leaves a descriptor at its first
      // slot there -- i | low << 8 | plen << 18 | (seeds of its run before the window) << 24 -- and a wave max-scan over
      // "slot+1 where a descriptor sits" tells every slot where its entry's run begins.  The lane of a slot then finds its
      // partner: the k-th T2 entry of the run, or, where mask bytes / strands / the entry itself drop members of the run,
      // the k-th one that stays.
      const bool plain = (MODE != MODE_FLIP) && !A.soft_mask;
      WSYNC();                                                   // the match's key reads are done: the array becomes the window
      for (int wb = 0; wb < T; wb += EWIN)
        { const int wn = T - wb < EWIN ? T - wb : EWIN;          // slots of this window
          for (int x = lane; 4*x < wn; x += 64)
            ((uint4 *) own32)[x] = make_uint4(0,0,0,0);
          WSYNC();
          { int o = off;
            #pragma unroll
            for (int r = 0; r < 4; r++)
              { const int cnt = (int) (res[r] >> 24);
                if (cnt > 0 && o + cnt > wb && o < wb + EWIN)
                  { const int before = o < wb ? wb - o : 0;
                    own32[o + before - wb] = (res[r] & 0xffffffu) | ((uint32_t) before << 24);
                  }
                o += cnt;
              }
          }
          WSYNC();
          int carry = 0;
          for (int s0 = 0; s0 < wn; s0 += 64)
            { const int slot = s0 + lane;                       // within the window
              int v = (slot < wn && own32[slot] != 0) ? slot+1 : 0;        // a descriptor is never 0: plen >= 12
              v = wave_incl_scan_max_dpp(v);
              v = v > carry ? v : carry;
              carry = __builtin_amdgcn_readlane(v,63);
              if (slot < wn)
                { const int start = v-1;
                  const uint32_t d = own32[start];
                  const int i = (int) (d & 0xff), plen = (int) ((d >> 18) & 0x3f);
                  int k = (slot - start) + (int) (d >> 24);
                  int j = (int) ((d >> 8) & 0x3ff);
                  if (plain)
                    { j += k;
                      if (MODE == MODE_SELF && j >= i)
                        j += 1;
                    }
                  else
                    { const int mlen = A.soft_mask ? plen : 41;
                      for (;; j++)
                        { if (A.soft_mask && (int) mB[j] >= mlen)
                            continue;
                          if (MODE == MODE_FLIP && (lds_c(cB,cw2,j) & A.sign2))
                            continue;
                          if (MODE == MODE_SELF && j == i)
                            continue;
                          if (k == 0)
                            break;
                          k -= 1;
                        }
                    }
                  const uint32_t spos = (MODE == MODE_SELF) ? pB[i] : pA[i];
                  const uint32_t sc = (MODE == MODE_SELF) ? lds_c(cB,cw2,i) : lds_c(cA,cw1,i);
                  const uint32_t cpos = pB[j], cc = lds_c(cB,cw2,j);
                  const uint32_t ssign = (sc & A.sign1) != 0, csign = (cc & A.sign2) != 0;
                  const int64_t gs = (int64_t) wb + slot;
                  const int64_t at = (gs < rem) ? O.chunk_pos + gs : nbase + (gs - rem);
                  if (at < A.cap)
                    A.out[at] = make_seed<MODE>(plen,spos,sc & (A.sign1-1),ssign,cpos,cc & (A.sign2-1),csign);
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
  return na;
}

template <int MODE, int T2CAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(T2CAP == 256 ? WAVE_OCC : 2,T2CAP == 256 ? WAVE_OCC : 2)))
void seed_merge_walk_kernel(merge_args A)
{ __shared__ __attribute__((aligned(16))) tile_lds<T2CAP> S;
  extern __shared__ __attribute__((aligned(16))) uint8_t cdyn[];     // contig|sign words of both sides: (T2CAP+8) cw2 + (T1CAP+8) cw1 bytes

  const int lane = threadIdx.x;
  const uint32_t *idx1 = A.v1.idx, *idx2 = (MODE == MODE_SELF) ? A.v1.idx : A.v2.idx;
  const int margin = A.freq + 2;
  walk_out O;
  O.chunk_pos = O.chunk_end = 0; O.tsum = 0;

  for (;;)
    { int r = 0;
      if (lane == 0)
        r = atomicAdd(A.next,1);
      r = __builtin_amdgcn_readfirstlane(r);
      if (r >= A.nranges)
        break;
      int p = (int) A.cuts[r];
      const int pe = (int) A.cuts[r+1];
      if (p >= pe)
        continue;
      uint32_t a = p > 0 ? idx1[p-1] : 0u, b = p > 0 ? idx2[p-1] : 0u;
      // Index entries of the next 64 prefixes, one per lane (clamped at the range end).  They are fetched one tile
      // ahead and travel HBM -> LDS like the tile data, not into registers: a register result would make the compiler
      // wait for ALL outstanding vector-memory operations where the loop uses it, i.e. for the acknowledgements of the
      // seed stores the tile before has just issued.  Their arrival is covered by the tile's own wait for its data
      // (issued after them, completed in order); the reads below are opaque to the compiler for the same reason.
#define IDX_ISSUE(P0)                                                                                                \
      { const int64_t e_ = (P0) + (lane < pe-(P0) ? lane : pe-(P0)-1);                                                  \
        G2L(idx1 + e_,&S.ixs[0][lane],4);                                                                              \
        if (MODE != MODE_SELF)                                                                                         \
          G2L(idx2 + e_,&S.ixs[1][lane],4);                                                                            \
      }
      IDX_ISSUE(p)
      VM_WAIT();
      while (p < pe)
        { uint32_t ca, cb;
          { const uint32_t at = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) &S.ixs[0][lane];
            uint64_t v;
            asm volatile("ds_read2st64_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(at) : "memory");
            ca = (uint32_t) v;
            cb = (MODE == MODE_SELF) ? ca : (uint32_t) (v >> 32);
          }
          const int navail = pe - p < XPC ? pe - p : XPC;
          // whole panels that fit a tile; a tile stays inside one block of 256 prefixes (key order = k-mer order)
          const bool fits = lane < navail && (ca - a) <= (uint32_t) T1CAP && (cb - b) <= (uint32_t) (MODE == MODE_SELF ? T1CAP : T2CAP)
                            && ((p + lane) >> 8) == (p >> 8);
          const int q = __popcll(__builtin_amdgcn_ballot_w64(fits));     // the conditions are monotone in the lane: a prefix mask
          const int adv = q > 0 ? q : 1;
          const uint32_t a1 = (uint32_t) __builtin_amdgcn_readlane((int) ca,adv-1);
          const uint32_t b1 = (uint32_t) __builtin_amdgcn_readlane((int) cb,adv-1);
          const int64_t n1 = (int64_t) a1 - a, n2 = (int64_t) b1 - b;
          // the index entries of the tile after this one are on their way while this one is processed
          const int pn = p + adv;
          if (pn < pe)
            IDX_ISSUE(pn)
          bool tiled = false;
          if (n1 > 0 && n2 > 0)
            { int lbl;
              if (q > 0)
                { walk_tile<MODE,T2CAP>(A,S,cdyn,a,(int) n1,b,(int) n2,0,(int) n1,false,margin,O,lbl);
                  tiled = true;
                }
              else if (MODE == MODE_SELF)
                { // one oversize panel against itself: runs of T1 entries with `margin` neighbours either side
                  const int C = T1CAP - 2*margin;
                  for (int64_t i0 = 0; i0 < n1; i0 += C)
                    { const int64_t i1 = i0 + C < n1 ? i0 + C : n1;
                      const int64_t s0 = i0 - margin > 0 ? i0 - margin : 0, s1 = i1 + margin < n1 ? i1 + margin : n1;
                      walk_tile<MODE,T2CAP>(A,S,cdyn,a+s0,(int) (s1-s0),a+s0,(int) (s1-s0),(int) (i0-s0),(int) (i1-s0),
                                            false,margin,O,lbl);
                    }
                  tiled = true;
                }
              else
                { // one oversize panel, streamed: a window of the T2 panel, the T1 entries it can finish, move on
                  int64_t aw = a, bw = b;
                  const int64_t ae = a1, be = b1;
                  while (aw < ae)
                    { const int n1w = (int) (ae - aw < T1CAP ? ae - aw : T1CAP);
                      const int n2w = (int) (be - bw < T2CAP ? be - bw : T2CAP);
                      const bool limit = bw + n2w < be;
                      const int na = walk_tile<MODE,T2CAP>(A,S,cdyn,aw,n1w,bw,n2w,0,n1w,limit,margin,O,lbl);
                      if (na == 0)
                        { // no T1 key within reach of this window: find where the next one lands
                          const uint64_t kq = A.v1.K[aw];
                          int64_t l = wave_lower_bound(A.v2.K,bw + n2w - margin,be,kq) - margin;
                          if (l <= bw) l = bw + 1;
                          bw = l;
                          if (bw >= be) break;                    // cannot happen: the last window has no limit
                          continue;
                        }
                      aw += na;
                      int64_t nbw = bw + lbl - margin;
                      if (nbw > bw) bw = nbw;
                    }
                  tiled = true;
                }
            }
          if (!tiled)                                          // no tile, no wait of a tile: the index entries may be on their way
            VM_WAIT();
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
  unsigned long long tsum = O.tsum;
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0 && tsum != 0)
    atomicAdd(A.tseed,tsum);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <int T2CAP>
static void launch_walk(int mode, int grid, size_t dyn, hipStream_t st, const merge_args &A)
{ if (mode == MODE_SELF)      hipLaunchKernelGGL((seed_merge_walk_kernel<MODE_SELF,T2CAP>),dim3(grid),dim3(64),dyn,st,A);
  else if (mode == MODE_FLIP) hipLaunchKernelGGL((seed_merge_walk_kernel<MODE_FLIP,T2CAP>),dim3(grid),dim3(64),dyn,st,A);
  else                        hipLaunchKernelGGL((seed_merge_walk_kernel<MODE_PAIR,T2CAP>),dim3(grid),dim3(64),dyn,st,A);
}

static int merge_impl(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                      const fga_merge_params *prm, int64_t capacity, fga_dseeds **out, fga_dseeds *append)
{ if (out != NULL) *out = NULL;
  if (dev == NULL || t1 == NULL || prm == NULL)
    { fga_set_error("fga_seed_merge: null argument");
      return 1;
    }
  const int self = (t2 == NULL);
  if (self) t2 = t1;
  if (self && prm->flip)
    { fga_set_error("fga_seed_merge: flip is meaningless for a self comparison");
      return 1;
    }
  if (t1->view.K == NULL || t2->view.K == NULL)
    { fga_set_error("fga_seed_merge: the index has no device view");
      return 1;
    }
  if (prm->freq < 1 || prm->freq > 255)
    { fga_set_error("fga_seed_merge: frequency cutoff must be in [1,255]");
      return 1;
    }
  // an index read from the pre-v1.3 layout holds no k-mer above the cutoff it was built with (FastGA.c:4959-4974)
  { const fga_dgix *ts[2] = { t1, t2 };
    for (int q = 0; q < 2; q++)
      if (ts[q]->legacy_cutoff > 0 && ts[q]->legacy_cutoff < prm->freq)
        { fga_set_error("genome index %d was built with a frequency cutoff of %d < the requested cutoff %d",
                        q+1,ts[q]->legacy_cutoff,prm->freq);
          return 1;
        }
  }
  FGA_HIP(hipSetDevice(dev->device));
  const int mode = self ? MODE_SELF : (prm->flip ? MODE_FLIP : MODE_PAIR);
  if (mode == MODE_PAIR && fga_dgix_make_forward(dev,(fga_dgix *) t1))      // first use as table 1 of a pair comparison
    return 1;

  merge_args A;
  memset(&A,0,sizeof(A));
  A.v1 = (mode == MODE_PAIR) ? t1->fview : t1->view;
  A.v2 = t2->view;
  A.sign1 = 0x80u << (8*(t1->contbytes-1)); A.sign2 = 0x80u << (8*(t2->contbytes-1));
  A.freq = prm->freq; A.soft_mask = prm->soft_mask;
  // prefix range: (0,0) = everything; an empty range elsewhere is an empty shard (prefix cuts of a low-complexity input)
  int64_t pb = prm->prefix_begin, pe = prm->prefix_end;
  if (pb < 0) pb = 0;
  if (pe > FGA_NPREFIX) pe = FGA_NPREFIX;
  if (pb == 0 && pe <= 0) pe = FGA_NPREFIX;
  const bool empty = pb >= pe;
  A.pbeg = (int) pb; A.pend = (int) (empty ? pb : pe);

  // cost at both ends of the prefix range (4 tiny D2H copies)
  uint32_t c1e = 0, c2e = 0, c1b = 0, c2b = 0;
  if (!empty)
    { FGA_HIP(hipMemcpy(&c1e,A.v1.idx + (A.pend-1),4,hipMemcpyDeviceToHost));
      FGA_HIP(hipMemcpy(&c2e,A.v2.idx + (A.pend-1),4,hipMemcpyDeviceToHost));
      if (A.pbeg > 0)
        { FGA_HIP(hipMemcpy(&c1b,A.v1.idx + (A.pbeg-1),4,hipMemcpyDeviceToHost));
          FGA_HIP(hipMemcpy(&c2b,A.v2.idx + (A.pbeg-1),4,hipMemcpyDeviceToHost));
        }
    }
  A.base = (int64_t) c1b + c2b + 2*(int64_t) A.pbeg;
  const int64_t total = ((int64_t) c1e + c2e + 2*(int64_t) A.pend) - A.base;

  fga_dseeds *S = append;
  if (S == NULL)
    { S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
      if (S == NULL)
        { fga_set_error("out of memory");
          return 1;
        }
      S->dev = dev;
      if (capacity <= 0)
        capacity = (mode == MODE_PAIR ? 4 : 2)*((int64_t) c1e - c1b) + (1<<20);      // two seeds per table-1 entry (the forward view holds half)
      S->capacity = capacity;
      // every wavefront of the kernel may leave up to a block unused, in this call and in a later append (-S)
      S->phys_capacity = capacity + 2*(int64_t) dev->ncu * 32 * FGA_SEED_BLOCK;
    }
  else
    capacity = S->capacity;
  const int64_t phys = S->phys_capacity;

  unsigned long long *counters = NULL;
  hipError_t err = hipSuccess;
  if (append != NULL)
    { counters = (unsigned long long *) S->dcount;
      if (S->valid == NULL)
        { fga_set_error("fga_seed_merge_append: the seed buffer was not produced by fga_seed_merge");
          return 1;
        }
    }
  else
    { const int64_t nb = (S->phys_capacity + FGA_SEED_BLOCK-1) / FGA_SEED_BLOCK + 2;
      if ((err = hipMalloc(&counters,8*sizeof(unsigned long long))) != hipSuccess ||
          (S->seeds = (fga_seed *) fga_dev_acquire(dev,SLOT_SEEDS,sizeof(fga_seed)*(size_t) S->phys_capacity)) == NULL ||
          (S->valid = (uint16_t *) fga_dev_acquire(dev,SLOT_VALID,sizeof(uint16_t)*(size_t) nb)) == NULL)
        { fga_set_error("fga_seed_merge: device allocation failed: %s",hipGetErrorString(err));
          hipFree(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); free(S);
          return 1;
        }
      S->slot = SLOT_SEEDS;
      S->dcount = (int64_t *) counters;
      hipMemsetAsync(counters,0,8*sizeof(unsigned long long),dev->stream);
      hipMemsetD16Async((hipDeviceptr_t) S->valid,(unsigned short) FGA_SEED_BLOCK,(size_t) nb,dev->stream);
    }
  A.out = S->seeds; A.cap = phys;
  A.count = counters; A.tseed = counters+1; A.hslots = counters+2;
  A.valid = S->valid; A.nblocks = (S->phys_capacity + FGA_SEED_BLOCK-1) / FGA_SEED_BLOCK + 2;

  int rc = 1;
  int64_t hslots = 0;
  unsigned long long hc[3] = {0,0,0};
  hipEvent_t ev2 = NULL;
  void *work = NULL;
  hipEventCreate(&ev2);
  dev->last_ms[FGA_STAGE_MERGE] = dev->last_ms[FGA_STAGE_MERGE_PARTITION] = 0.f;
  if (!empty)
    { // the sub-tile margin FREQ+2 must leave room in a window: the wide-window build takes over for large cutoffs
      const bool wide = 2*(prm->freq + 2) > 256 - 64;
      const int t2cap = wide ? 1024 : 256;
      const size_t dyn = (size_t) (t2cap + 8)*A.v2.cw + (size_t) (T1CAP + 8)*A.v1.cw + 16;
      // every wavefront of the launch is resident (they are persistent): as many as the LDS of a CU holds, at most the
      // register budget's
      const size_t lds = ((wide ? sizeof(tile_lds<1024>) : sizeof(tile_lds<256>)) + dyn + 511) / 512 * 512;
      int per_cu = (int) ((160*1024) / lds);
      const int fit = per_cu;
      const int occ = wide ? 8 : 4*WAVE_OCC;
      if (per_cu > occ) per_cu = occ;
      { const char *ev = getenv("FGA_MERGE_WAVES");
        if (ev != NULL && atoi(ev) > 0 && atoi(ev) <= fit && atoi(ev) <= 28) per_cu = atoi(ev);     // phys_capacity's slack covers 32 per CU
      }
      int grid = dev->ncu * per_cu;
      // ranges of equal merge cost, a few per wavefront, taken off a queue
      int nranges = grid*RANGES_PER_WAVE;
      if ((int64_t) nranges > total/2048 + 1) nranges = (int) (total/2048) + 1;
      if (grid > nranges) grid = nranges;
      work = fga_dev_acquire(dev,SLOT_TILES,sizeof(int64_t)*(size_t) (nranges+2) + 64);
      if (work == NULL)
        { fga_set_error("fga_seed_merge: device allocation failed");
          goto done;
        }
      int64_t *cuts = (int64_t *) work;
      int *qhead = (int *) (counters + 4);
      A.cuts = cuts; A.nranges = nranges; A.next = qhead;
      hipMemsetAsync(qhead,0,sizeof(unsigned long long),dev->stream);
      hipEventRecord(dev->ev0,dev->stream);
      hipLaunchKernelGGL(range_cut_kernel,dim3((nranges+1+255)/256),dim3(256),0,dev->stream,
                         A.v1.idx,A.v2.idx,A.pbeg,A.pend,A.base,total,nranges,cuts);
      hipEventRecord(dev->ev1,dev->stream);
      if (wide) launch_walk<1024>(mode,grid,dyn,dev->stream,A);
      else      launch_walk<256>(mode,grid,dyn,dev->stream,A);
      hipEventRecord(ev2,dev->stream);
    }
  // one round trip: the three counters
  { unsigned long long *pin = (unsigned long long *) fga_dev_pinned(dev,sizeof(unsigned long long)*8);
    if (pin == NULL)
      { fga_set_error("fga_seed_merge: pinned staging allocation failed");
        goto done;
      }
    err = hipMemcpyAsync(pin,counters,3*sizeof(unsigned long long),hipMemcpyDeviceToHost,dev->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(dev->stream);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess)
      { fga_set_error("fga_seed_merge: kernel failed: %s",hipGetErrorString(err));
        goto done;
      }
    hc[0] = pin[0]; hc[1] = pin[1]; hc[2] = pin[2];
    hslots = (int64_t) hc[2];
  }
  if (!empty)
    { hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE_PARTITION],dev->ev0,dev->ev1);
      // the merge stage = everything the launch runs: range cuts and the walk kernel
      hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE],dev->ev0,ev2);
    }
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
  if (S->phys_count > S->phys_capacity)
    S->count = S->phys_count;              // overflow of the block-allocated buffer: an upper bound of what is needed
  if (S->count > S->capacity || S->phys_count > S->phys_capacity)
    { fga_set_error("fga_seed_merge: %lld seeds exceed the buffer capacity %lld (re-run with a larger capacity)",
                    (long long) S->count,(long long) S->capacity);
      return 2;
    }
  return 0;
}

// cuts[0..nshards]: 12-mer prefix ranges [cuts[r], cuts[r+1]) of equal merge cost (entries of both tables + prefixes), the
// phase-1 shards of a multi-GPU run -- the reference splits its merge threads the same way (FastGA.c:2291-2321)
__global__ void prefix_cut_kernel(const uint32_t *idx1, const uint32_t *idx2, int64_t total, int nshards, int64_t *cuts)
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
          const int64_t c = (int64_t) idx1[mid] + (int64_t) idx2[mid] + 2*((int64_t) mid+1);
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
  if (t1->view.idx == NULL || t2->view.idx == NULL)
    { fga_set_error("fga_merge_prefix_cuts: the index has no device view");
      return 1;
    }
  FGA_HIP(hipSetDevice(dev->device));
  const int64_t total = t1->nents + t2->nents + 2*(int64_t) FGA_NPREFIX;
  int64_t *d = (int64_t *) fga_dev_acquire(dev,SLOT_MISC,sizeof(int64_t)*(size_t) (nshards+1));
  if (d == NULL)
    { fga_set_error("fga_merge_prefix_cuts: device allocation failed");
      return 1;
    }
  hipLaunchKernelGGL(prefix_cut_kernel,dim3((nshards+1+63)/64),dim3(64),0,dev->stream,t1->view.idx,t2->view.idx,total,nshards,d);
  hipError_t e = hipMemcpyAsync(cuts,d,sizeof(int64_t)*(size_t) (nshards+1),hipMemcpyDeviceToHost,dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  fga_dev_release(dev,SLOT_MISC,d);
  if (e != hipSuccess)
    { fga_set_error("fga_merge_prefix_cuts: %s",hipGetErrorString(e));
      return 1;
    }
  // non-decreasing, and no empty shard at prefix 0: the range (0,0) means "everything" to fga_seed_merge
  for (int w = 1; w <= nshards; w++)
    { if (cuts[w] < cuts[w-1]) cuts[w] = cuts[w-1];
      if (cuts[w] < 1) cuts[w] = 1;
    }
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
