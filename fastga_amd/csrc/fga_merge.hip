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

#ifndef T1CAP
#define T1CAP           256                  // table-1 entries per tile: four rounds of one entry per lane
#endif
#define FGA_MERGE_MAX_FREQ 1982               // 2 x (cutoff + 2) entries of margin and 128 to work on must fit the largest window (4096)
#ifndef T2STD
#define T2STD           512                  // table-2 entries per tile / window of the standard build
#endif
#define XPC             128                  // prefixes per tile at most (two index entries per lane)
#define EWIN            1024                 // slots per emission window (descriptor dwords in the key array)
#ifndef WAVE_OCC
// The words the wavefronts of a launch update with atomics live 256 bytes apart: atomics on ONE cache line are served one
// after the other, ~88 per microsecond for the whole chip (MI355X_MICROARCH.md), whichever words of the line they name.
// Until this was measured the slot counter, the statistics and the eight queue heads shared one 64-byte line: 90,000
// atomics per launch on the 100 Mbp pair = 1.0 of its 1.25 ms.  (In 8-byte / 4-byte units:)
#define CTR_COUNT   0                        // slots handed out
#define CTR_STATS   32                       // sum of plen, slots left unused
#define CTR_QUEUE   64                       // queue head k at CTR_QUEUE + 32 k
#define CTR_COLD    320                      // the end-of-kernel arguments (read only)
#define CTR_WORDS   384
#define QSTRIDE     64                       // ints between two queue heads
#define WAVE_OCC        4                    // resident wavefronts per SIMD the register budget is held to
#endif
#ifndef RANGES_PER_WAVE
#define RANGES_PER_WAVE 4
#endif

#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL,"wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define VM_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70)      // vmcnt(0), the other counters left alone

struct merge_cold                     // what a wavefront needs once, at its end (read from memory, not held in registers)
  { unsigned long long *tseed;        // sum of plen (the reference's "ave. len" statistic)
    unsigned long long *hslots;       // slots left unused
    uint16_t *valid;                  // seeds per 1024-slot block of `out` (holes are left open, consumers skip them)
    int64_t   nblocks;
  };

struct merge_args
  { const uint64_t *K1; const uint32_t *P1; const uint8_t *C1; const uint8_t *M1; const uint32_t *idx1;
    const uint64_t *K2; const uint32_t *P2; const uint8_t *C2; const uint8_t *M2; const uint32_t *idx2;
    fga_car car1, car2;                 // the prefix indices hold low words: where their counts pass multiples of 2^32 (fga_device.hpp)
    const uint8_t  *L2;
    int   cw1, cw2;                   // bytes per contig|sign word
    uint32_t sign1, sign2;            // sign bit of the contig words
    int   freq, soft_mask;
    fga_seed *out; int64_t cap;
    unsigned long long *count;        // slots handed out
    const int64_t *cuts;              // [nranges+1] prefix boundaries
    int       nranges;
    int      *next;                   // [8] queue heads, one per XCD
    const merge_cold *cold;
  };

// ranges of the prefix space for the wavefronts' queues: `nbig` ranges of equal cost over the first five eighths of the
// work, then `nranges - nbig` small ones over the rest (the queues are taken in order, so the launch ends on small
// pieces: its tail is one small range, not a quarter of a wavefront's share)
// (One wavefront per cut probing 64 split points a round -- four rounds instead of a lane's 24 dependent loads -- was tried:
// ten times the memory transactions, 50 us instead of 19.)
__global__ void range_cut_kernel(const uint32_t *idx1, const uint32_t *idx2, fga_car car1, fga_car car2, int pbeg, int pend, int64_t base,
                                 int64_t total, int nranges, int nbig, int64_t *cuts)
{ const int w = blockIdx.x*blockDim.x + threadIdx.x;
  if (w > nranges)
    return;
  int64_t p = pbeg;
  if (w == nranges)
    p = pend;
  else if (w > 0)
    { const int64_t t34 = (total >> 1) + (total >> 3);
      const int64_t target = base + (w <= nbig ? (t34 / nbig) * w : t34 + ((total - t34) / (nranges - nbig)) * (w - nbig));
      int lo = pbeg, hi = pend;
      while (lo < hi)
        { const int mid = lo + ((hi-lo) >> 1);
          const int64_t c = fga_idx_abs(idx1,car1,mid) + fga_idx_abs(idx2,car2,mid) + 2*((int64_t) mid+1);
          if (c > target) hi = mid; else lo = mid+1;
        }
      p = lo;
    }
  cuts[w] = p;
}

// LCP in bases of two keys of one panel (same prefix byte above the 56-bit suffix): 12 .. 40
__device__ __forceinline__ int lcp_key(uint64_t a, uint64_t b)
{ const uint64_t x = a ^ b;
  return x == 0 ? 40 : 8 + (__clzll((long long) x) >> 1);       // clz >= 8: 12 + (clz - 8)/2
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

// smallest j in [lo,hi] with K[j] >= kq (one panel), all 64 lanes probing HBM: the range shrinks 64-fold a
// round (only when a window of an oversize panel holds no T1 key)
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

#ifdef MERGE_PROF      // per-phase cycle accounting of every wavefront (tools/merge_bench.py prints it): timing builds only
__device__ unsigned long long merge_prof[16*32];    // counter k at [32 k]: a cache line each -- 3,072 wavefronts flushing 16 counters of ONE line
                                                     // at their ends (~88 atomics per microsecond and line) stalled that memory channel
                                                     // for everybody and made the launch's last ranges take five times as long
__device__ unsigned long long merge_tl[8192*4];     // per workgroup: start, end (100 MHz clock), tiles, ranges
static int merge_prof_grid = 0;
#define XPROF(k)   { unsigned long long _n = clock64(); O.pa[k] += _n - O.pt; O.pt = _n; }
#else
#define XPROF(k)
#endif

struct walk_out                      // a wavefront's current output chunk and statistics (wave-uniform)
  { int64_t chunk_pos, chunk_end;
    unsigned long long tsum;
#ifdef MERGE_PROF
    unsigned long long pt, pa[16], t0, ntile, nrange;
#endif
  };

// the LDS of one wavefront.  Every array is filled in 16-byte pieces from a 16-byte aligned address of its HBM array, so
// entry 0 of a window sits (window start mod 2 / 4 / 16) entries into it.
template <int T2CAP>
struct tile_lds
  { uint64_t keyB0[T2CAP + 4];       // T2 keys; emission descriptors later
    uint32_t pB0[T2CAP + 8];         // T2 positions
    uint32_t pA0[T1CAP + 8];         // T1 positions
    uint8_t  lcpB0[T2CAP + 32];      // T2 lcp bytes
    uint8_t  mB0[T2CAP + 32];        // T2 mask bytes  (soft mask runs only)
    uint8_t  mA0[T1CAP + 32];        // T1 mask bytes
    uint32_t ixs[2][2][XPC + 4];     // [buffer][table]: index entries of the next XPC prefixes from slot 4 on, slot 3 = the entry before
  };

__device__ __forceinline__ uint32_t lds_c(const uint8_t *c, int cw, int i)
{ if (cw == 1) return c[i];
  if (cw == 2) return ((const uint16_t *) c)[i];
  return ((const uint32_t *) c)[i];
}

// (cache policy: nt -- aux 2 -- on these loads and on the seed stores was measured in round 5: 0.789-0.799 ms against 0.789-0.810
// for the default policy on the bench pair, inside the run-to-run spread; the default stays)
#define G2L(gp,lp,sz) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (gp), \
                                                       (__attribute__((address_space(3))) void *) (lp),sz,0,0)

// nbytes from the (wave-uniform, 16-byte aligned) HBM address g to the LDS address l, 16 bytes per lane and instruction
__device__ __forceinline__ void lds_fill(const void *g, void *l, int nbytes, int lane16)
{ const char *gc = (const char *) g;
  char *lc = (char *) l;
  int x0 = 0;
  // the LDS operand is the wavefront's base (the hardware adds lane x 16 itself); the HBM address is a uniform base plus a
  // 32-bit lane offset
  for (; x0 + 1024 <= nbytes; x0 += 1024)
    G2L(gc + x0 + (size_t) (uint32_t) lane16,lc + x0,16);
  if (x0 + lane16 < nbytes)
    G2L(gc + x0 + (size_t) (uint32_t) lane16,lc + x0,16);
}

// How the match's result for one table-1 entry (and the emission's descriptor made from it) is packed: entry i of the tile
// (self: i - low), first run member `low`, plen, number of seeds.  32 bits in the windows of up to 1024 entries (8 + 10 + 6 +
// 8 bits: run counts below 256, i.e. -f <= 255); 64 bits in the 4096-entry windows of the build for larger cutoffs.
template <int T2CAP, bool FLAG = false>
struct res_fmt
  { static constexpr bool BIG = (T2CAP > 1024);
    typedef typename std::conditional<BIG,uint64_t,uint32_t>::type word;
    // i: the entry's index in the tile (< T1CAP = 256), or in a self comparison its distance from the run's start (<= -f)
    // plen (12 .. 40) in six bits.  FLAG (the self and the FLIP kernels): one bit says that the run has members that do not
    // count (a mask byte, the strand) -- only then does the emission have to walk the run to find a seed's partner; in the
    // narrow word the bit is plen's sixth, and plen - 11 (never 0: a descriptor is told from an empty slot by that) takes five.
    // The plain pair kernel keeps the word without the flag (its emission never walks; the two instructions per entry showed).
    static constexpr int I_BITS = BIG ? 16 : 8, LOW_SH = I_BITS, LOW_BITS = BIG ? 16 : 10, PLEN_SH = LOW_SH + LOW_BITS,
                         PLEN_BITS = (BIG || !FLAG) ? 6 : 5, PLEN_OFF = (BIG || !FLAG) ? 0 : 11, DIRTY_SH = PLEN_SH + PLEN_BITS,
                         CNT_SH = BIG ? 40 : 24;
    static __device__ __forceinline__ word pack(int i, int low, int plen, int cnt, bool dirty)
    { return (word) (uint32_t) i | ((word) (uint32_t) low << LOW_SH) | ((word) (uint32_t) (plen - PLEN_OFF) << PLEN_SH)
             | (FLAG ? ((word) (dirty ? 1u : 0u) << DIRTY_SH) : (word) 0) | ((word) (uint32_t) cnt << CNT_SH); }
    static __device__ __forceinline__ bool dirty(word d) { return FLAG ? ((d >> DIRTY_SH) & 1) != 0 : true; }
    static __device__ __forceinline__ int  cnt(word r)   { return (int) (r >> CNT_SH); }
    static __device__ __forceinline__ word body(word r)  { return r & (((word) 1 << CNT_SH) - 1); }          // all but the count
    static __device__ __forceinline__ int  i(word d)     { return (int) (d & (((word) 1 << I_BITS) - 1)); }
    static __device__ __forceinline__ int  low(word d)   { return (int) ((d >> LOW_SH) & (((word) 1 << LOW_BITS) - 1)); }
    static __device__ __forceinline__ int  plen(word d)  { return (int) ((d >> PLEN_SH) & (((word) 1 << PLEN_BITS) - 1)) + PLEN_OFF; }
  };

// The match of up to NR rounds of T1 entries (entry c = r*64 + lane of the tile) side by side.  qe: the entry's panel
// among the tile's (index into ix2, whose slot -1 holds the entries before the tile), or < 0: one panel = the window.
template <int MODE, int NR, int T2CAP>
__device__ __forceinline__ void match_rounds(const merge_args &A, const uint64_t *keyB, const uint8_t *lcpB, const uint8_t *mA,
                                             const uint8_t *mB, const uint8_t *cB, const uint64_t *k1, const uint32_t *ix2,
                                             uint32_t b, int plo, bool panels, int n2, int na, int t_lo,
                                             typename res_fmt<T2CAP>::word *res, int &total, unsigned long long &tsum, int &lb_last,
                                             walk_out &O)
{ const int lane = threadIdx.x;
  const int freq = A.freq;
  uint64_t ks[NR];
  int base[NR], len[NR], pb0[NR], pb1[NR];
  bool any = false;
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { const int c = r*64 + lane;
      const int i = (MODE == MODE_SELF) ? (c < na ? t_lo + c : t_lo) : c;
      ks[r] = (MODE == MODE_SELF) ? keyB[i] : k1[r];
      if (panels)
        { const int qe = (int) (((uint32_t) (ks[r] >> 56) - (uint32_t) plo) & 0xff);
          pb0[r] = (int) (ix2[qe-1] - b); pb1[r] = (int) (ix2[qe] - b);      // entries before the panel, entries up to its end
        }
      else
        { pb0[r] = 0; pb1[r] = n2; }
      base[r] = pb0[r]; len[r] = pb1[r] - pb0[r];
      if (MODE == MODE_SELF) { base[r] = i; len[r] = 0; }
      any = any || len[r] > 1;
    }
  if (MODE != MODE_SELF)
    { // lower bound of every key inside its own panel: halving steps, the chains of all rounds in flight together
      while (__builtin_amdgcn_ballot_w64(any) != 0)
        { any = false;
          #pragma unroll
          for (int r = 0; r < NR; r++)
            { const int half = len[r] >> 1;
              const bool go = len[r] > 1;
              const uint64_t kv = keyB[base[r] + half - (go ? 1 : 0)];                  // read by every lane: the reads of all rounds in flight together
              const bool lt = go & (kv < ks[r]);                                        // keys of one panel: same prefix byte
              base[r] += lt ? half : 0;
              len[r]  -= go ? half : 0;
              any = any || len[r] > 1;
            }
        }
      uint64_t kf[NR];
      #pragma unroll
      for (int r = 0; r < NR; r++)
        kf[r] = keyB[base[r]];
      #pragma unroll
      for (int r = 0; r < NR; r++)
        base[r] += ((len[r] > 0) & (kf[r] < ks[r])) ? 1 : 0;
    }
  XPROF(8)
  uint64_t kb[NR], kc[NR];
  int bd[NR], bu[NR], nb[NR], na_[NR];
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { if (MODE == MODE_SELF) { nb[r] = base[r]-1; na_[r] = base[r]+1; }
      else                   { nb[r] = base[r]-1; na_[r] = base[r]; }
      kb[r] = keyB[nb[r] >= 0 ? nb[r] : 0]; kc[r] = keyB[na_[r]];
      bd[r] = (int) lcpB[nb[r] >= 0 ? nb[r] : 0]; bu[r] = (int) lcpB[na_[r]+1];
    }
#ifdef MERGE_PROF
  if (__builtin_amdgcn_ballot_w64(kb[0] + kc[0] + bd[0] + bu[0] == 0x1234567) != 0) tsum += 1;     // the reads have landed
#endif
  XPROF(9)
  // The run of every entry, grown on the table's own lcp bytes (lcpB[x] = what entry x shares with entry x-1) as the reference
  // grows it one entry at a time -- down while low > panel start, the run spans at most freq + 1 entries below the lower bound
  // and the next byte reaches plen; then up likewise.  The first step of either direction is decided on the bytes read above;
  // the rest takes EIGHT lcp bytes per LDS read (one unaligned ds_read_b64, the bytes compared side by side), all rounds'
  // reads in flight together: a tile pays two or three dependent LDS round trips for its longest run instead of one per
  // member and round (the entries of repeat families -- every tile has some -- grow to the cutoff: 11 steps either way).
  total = 0;
  int low[NR], hgh[NR], lbnd[NR], plen[NR], ei[NR];
  bool ok[NR], cd[NR], cu[NR];
  bool anyd = false;
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { const int c = r*64 + lane;
      const bool act = c < na;
      ei[r] = (MODE == MODE_SELF) ? base[r] : c;
      if (MODE == MODE_SELF) { low[r] = ei[r]; hgh[r] = ei[r]+1; lbnd[r] = ei[r]; }
      else                   { low[r] = hgh[r] = lbnd[r] = base[r]; }
      const bool hasb = nb[r] >= pb0[r], hasa = na_[r] < pb1[r];
      const int lkb = hasb ? lcp_key(ks[r],kb[r]) : 0, lka = hasa ? lcp_key(ks[r],kc[r]) : 0;
      plen[r] = lkb > lka ? lkb : lka;                      // 0: no T2 entry of this panel next to the key
      ok[r] = act && plen[r] >= 12;
      const bool gd = ok[r] && lkb >= plen[r];
      low[r] -= gd ? 1 : 0;
      cd[r] = gd && low[r] > pb0[r] && lbnd[r]-low[r] <= freq && bd[r] >= plen[r];
      low[r] -= cd[r] ? 1 : 0;
      cu[r] = ok[r] && lka >= plen[r];                      // (the upward growth's first test; completed below, with the final low)
      anyd = anyd || cd[r];
    }
  while (__builtin_amdgcn_ballot_w64(anyd) != 0)
    { uint64_t w[NR];
      #pragma unroll
      for (int r = 0; r < NR; r++)
        __builtin_memcpy(&w[r],lcpB + (cd[r] ? low[r] : 7) - 7,8);          // bytes low-7 .. low: the top byte is lcpB[low]
      anyd = false;
      #pragma unroll
      for (int r = 0; r < NR; r++)
        { const uint32_t k = (uint32_t) plen[r] * 0x01010101u;
          const uint32_t nl = ((((uint32_t) w[r] | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;             // 0x80 where a byte is below plen
          const uint32_t nh = ((((uint32_t) (w[r] >> 32) | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;
          const int t = nh ? (__builtin_clz(nh) >> 3) : 4 + (nl ? (__builtin_clz(nl) >> 3) : 4);
          int lim = low[r] - pb0[r];
          const int l2 = freq + 1 - (lbnd[r] - low[r]);
          lim = lim < l2 ? lim : l2;
          int st = t < lim ? t : lim;
          st = (cd[r] && st > 0) ? st : 0;
          low[r] -= st;
          cd[r] = st == 8;
          anyd = anyd || cd[r];
        }
    }
  bool anyu = false;
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { const bool gu = cu[r] && hgh[r] < pb1[r] && hgh[r]-low[r] <= freq;
      hgh[r] += gu ? 1 : 0;
      cu[r] = gu && hgh[r] < pb1[r] && hgh[r]-low[r] <= freq && bu[r] >= plen[r];
      hgh[r] += cu[r] ? 1 : 0;
      anyu = anyu || cu[r];
    }
  while (__builtin_amdgcn_ballot_w64(anyu) != 0)
    { uint64_t w[NR];
      #pragma unroll
      for (int r = 0; r < NR; r++)
        __builtin_memcpy(&w[r],lcpB + (cu[r] ? hgh[r] : 0),8);              // bytes hgh .. hgh+7
      anyu = false;
      #pragma unroll
      for (int r = 0; r < NR; r++)
        { const uint32_t k = (uint32_t) plen[r] * 0x01010101u;
          const uint32_t nl = ((((uint32_t) w[r] | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;
          const uint32_t nh = ((((uint32_t) (w[r] >> 32) | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;
          const int t = nl ? (__builtin_ctz(nl) >> 3) : 4 + (nh ? (__builtin_ctz(nh) >> 3) : 4);
          int lim = pb1[r] - hgh[r];
          const int l2 = freq + 1 - (hgh[r] - low[r]);
          lim = lim < l2 ? lim : l2;
          int st = t < lim ? t : lim;
          st = (cu[r] && st > 0) ? st : 0;
          hgh[r] += st;
          cu[r] = st == 8;
          anyu = anyu || cu[r];
        }
    }
  #pragma unroll
  for (int r = 0; r < NR; r++)
    { const int i = ei[r];
      const int mlen = A.soft_mask ? plen[r] : 41;
      bool pass = ok[r] && hgh[r]-low[r] < freq;
      if (A.soft_mask)
        pass = pass && (int) (MODE == MODE_SELF ? mB[i] : mA[i]) < mlen;
      int cnt;
      if (MODE != MODE_FLIP && A.soft_mask)
        { // members whose mask byte is below mlen, eight bytes per LDS read (the entry itself, whose byte `pass` has tested,
          // is among them in a self comparison)
          cnt = 0;
          if (pass)
            { const uint32_t k = (uint32_t) mlen * 0x01010101u;
              for (int j = low[r]; j < hgh[r]; j += 8)
                { uint64_t w;
                  __builtin_memcpy(&w,mB + j,8);
                  const int n = hgh[r] - j < 8 ? hgh[r] - j : 8;
                  const uint64_t keep = n < 8 ? ((1ull << (8*n)) - 1ull) : ~0ull;
                  const uint32_t bl = ((((uint32_t) w | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;           // 0x80 where a byte is below mlen
                  const uint32_t bh = ((((uint32_t) (w >> 32) | 0x80808080u) - k) & 0x80808080u) ^ 0x80808080u;
                  cnt += __popc(bl & (uint32_t) keep) + __popc(bh & (uint32_t) (keep >> 32));
                }
              cnt -= (MODE == MODE_SELF) ? 1 : 0;
            }
        }
      else if (MODE == MODE_FLIP)
        { cnt = 0;
          if (pass)
            for (int j = low[r]; j < hgh[r]; j++)
              { if (A.soft_mask && (int) mB[j] >= mlen)
                  continue;
                if (MODE == MODE_FLIP && (lds_c(cB,A.cw2,j) & A.sign2))
                  continue;
                if (MODE == MODE_SELF && j == i)
                  continue;
                cnt += 1;
              }
        }
      else
        cnt = pass ? (hgh[r]-low[r]) - (MODE == MODE_SELF ? 1 : 0) : 0;
      res[r] = (pass && cnt > 0) ? res_fmt<T2CAP,MODE != MODE_PAIR>::pack(MODE == MODE_SELF ? i - low[r] : i,low[r],plen[r],cnt,
                                                        cnt != (hgh[r]-low[r]) - (MODE == MODE_SELF ? 1 : 0)) : 0;
      total += cnt;
      tsum += (unsigned long long) cnt * plen[r];
    }
  XPROF(10)
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

// One tile: T1 entries [a0, a0+n1) against the T2 window [b0, b0+n2) (self: one window, its entries [t_lo, t_hi) emit).
// panels: the window is made of whole panels, those of prefixes p0 .. whose index entries sit in ix1 / ix2 (slot -1: the
// entries before the tile); otherwise it is a stretch of ONE panel, and with `limit` (the stretch does not reach the panel's
// end) only the T1 entries whose lower bound is at most n2 - margin are consumed (a prefix of them).  Returns their
// number; lb_last = lower bound (window coordinates) of the last one.
template <int MODE, int T2CAP>
__device__ __forceinline__ int walk_tile(const merge_args &A, tile_lds<T2CAP> &S, uint8_t *cdyn, const uint32_t *ix1,
                                         const uint32_t *ix2, int p0, bool panels, int64_t a0, int n1,
                                         int64_t b0, int n2, int t_lo, int t_hi, bool limit, int margin, walk_out &O,
                                         int &lb_last)
{ const int lane = threadIdx.x;
  const int lane16 = lane*16;
  const int cw1 = A.cw1, cw2 = A.cw2;
  const int obk = (int) (b0 & 1), obp = (int) (b0 & 3), obc = (int) (b0 & 15), oap = (int) (a0 & 3), oac = (int) (a0 & 15);
  uint64_t *keyB = S.keyB0 + obk;
  uint32_t *pB = S.pB0 + obp, *pA = S.pA0 + oap;
  uint8_t  *lcpB = S.lcpB0 + obc, *mB = S.mB0 + obc, *mA = S.mA0 + oac;
  uint8_t  *cB0 = cdyn, *cA0 = cdyn + (size_t) (T2CAP + 32)*cw2;
  uint8_t  *cB = cB0 + (size_t) obc*cw2, *cA = cA0 + (size_t) oac*cw1;
  typename res_fmt<T2CAP>::word *ownd = (typename res_fmt<T2CAP>::word *) S.keyB0;      // the emission's descriptors reuse the key array (keys are done with by then)

  // 1. HBM -> LDS (global_load_lds): T2 keys, lcp bytes, payloads; T1 payloads.  T1 keys -> registers.
  lds_fill(A.K2 + (b0 - obk),S.keyB0,(n2 + obk)*8,lane16);
  lds_fill(A.P2 + (b0 - obp),S.pB0,(n2 + obp)*4,lane16);
  lds_fill(A.L2 + (b0 - obc),S.lcpB0,n2 + obc + 2,lane16);              // two lcp bytes beyond the window are read (never used)
  lds_fill(A.C2 + (size_t) (b0 - obc)*cw2,cB0,(n2 + obc)*cw2,lane16);
  if (A.soft_mask)
    lds_fill(A.M2 + (b0 - obc),S.mB0,n2 + obc,lane16);
  uint64_t k1[4];
  if (MODE != MODE_SELF)
    { lds_fill(A.P1 + (a0 - oap),S.pA0,(n1 + oap)*4,lane16);
      lds_fill(A.C1 + (size_t) (a0 - oac)*cw1,cA0,(n1 + oac)*cw1,lane16);
      if (A.soft_mask)
        lds_fill(A.M1 + (a0 - oac),S.mA0,n1 + oac,lane16);
      const uint64_t *kg = A.K1 + a0;
      #pragma unroll
      for (int r = 0; r < 4; r++)
        { const int i = r*64 + lane;
          k1[r] = kg[i < n1 ? i : n1-1];
        }
    }
  else
    { k1[0] = k1[1] = k1[2] = k1[3] = 0; }
  // the direct-to-LDS loads are asynchronous and nothing below depends on a VGPR they return: wait for them by hand
  XPROF(1)
#ifdef MERGE_PROF
  O.ntile += 1;
#endif
  VM_WAIT();
  WSYNC();
  XPROF(2)

  // 2. which T1 entries this window can finish
  int na = (MODE == MODE_SELF) ? t_hi - t_lo : n1;
  if (MODE != MODE_SELF && limit)
    { const uint64_t kl = keyB[n2 - margin];        // lower bound <= n2 - margin  <=>  key <= this one
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
#ifdef MERGE_FILL_ONLY                  // timing experiment: tiles are cut and filled, nothing else
  lb_last = n2 - margin - 1;
  WSYNC();
  return na;
#endif

  XPROF(3)
  // 3. match: result per round packed i (8 bits; self: i - low) | low | plen | seeds (res_fmt)
  typedef res_fmt<T2CAP,MODE != MODE_PAIR> RF;
  typename RF::word res[4];
  int total = 0;
  { const int nr = (na + 63) >> 6;
    const uint32_t b32 = (uint32_t) b0;
    const int plo = p0 & 0xff;
    const uint32_t *ixq = (MODE == MODE_SELF) ? ix1 : ix2;
    if (nr == 1)      match_rounds<MODE,1,T2CAP>(A,keyB,lcpB,mA,mB,cB,k1,ixq,b32,plo,panels,n2,na,t_lo,res,total,O.tsum,lb_last,O);
    else if (nr == 2) match_rounds<MODE,2,T2CAP>(A,keyB,lcpB,mA,mB,cB,k1,ixq,b32,plo,panels,n2,na,t_lo,res,total,O.tsum,lb_last,O);
    else if (nr == 3) match_rounds<MODE,3,T2CAP>(A,keyB,lcpB,mA,mB,cB,k1,ixq,b32,plo,panels,n2,na,t_lo,res,total,O.tsum,lb_last,O);
    else              match_rounds<MODE,4,T2CAP>(A,keyB,lcpB,mA,mB,cB,k1,ixq,b32,plo,panels,n2,na,t_lo,res,total,O.tsum,lb_last,O);
  }

  XPROF(4)
  // 4. slots and emission.  Most table-1 entries with seeds have exactly ONE (a k-mer met once in the other genome): such an
  // entry writes its seed itself -- its own position and contig word, its run's first member: four LDS reads and the store --
  // into the first `T1` slots of the tile's stretch (one per entry, in entry order); only the seeds of the other entries (runs of
  // several members, or runs in which mask bytes / strands / the entry itself drop members) go through the slot windows below
  const bool plain = (MODE != MODE_FLIP) && !A.soft_mask;
  int T, T1, off1, off;
  typename RF::word resm[4];
  { int n1s = 0, nm = 0;
    #pragma unroll
    for (int r = 0; r < 4; r++)
      { const int cnt = RF::cnt(res[r]);
        const bool single = cnt == 1 && (plain || !RF::dirty(res[r]));
        n1s += single ? 1 : 0;
        nm  += single ? 0 : cnt;
        resm[r] = single ? (typename RF::word) 0 : res[r];
      }
    if (RF::BIG)
      { off1 = wave_excl_scan_add_dpp(n1s,T1); off = wave_excl_scan_add_dpp(nm,T); }
    else                                 // (counts below 256: 64 lanes x 4 entries fit 16 bits) both sums in one scan
      { int tt;
        const int o = wave_excl_scan_add_dpp(nm | (n1s << 16),tt);
        off = o & 0xffff; off1 = o >> 16; T = tt & 0xffff; T1 = tt >> 16;
      }
    T += T1;                             // slots of the tile: [0, T1) the single seeds, [T1, T) the others
  }
#ifdef MERGE_NO_EMIT                    // timing experiment: matched, not emitted
  O.tsum += T; T = 0; T1 = 0;
#endif
  if (T > 0)
    { const int64_t rem = O.chunk_end - O.chunk_pos;
      int64_t nbase = 0, nsize = 0;
      if ((int64_t) T > rem)
        { nsize = (((int64_t) T - rem) + FGA_SEED_BLOCK-1) & ~(int64_t) (FGA_SEED_BLOCK-1);    // whole blocks, block aligned
          unsigned long long bb = 0;
          if (lane == 0)
            bb = atomicAdd(A.count,(unsigned long long) nsize);
          const uint32_t blo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) bb);
          const uint32_t bhi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (bb >> 32));
          nbase = (int64_t) (((uint64_t) bhi << 32) | blo);
        }
      XPROF(11)
      if (T1 > 0)                        // the single seeds, entry-parallel
        { int o1 = off1;
          #pragma unroll
          for (int r = 0; r < 4; r++)
            { const typename RF::word d = res[r];
              if (RF::cnt(d) == 1 && (plain || !RF::dirty(d)))
                { const int plen = RF::plen(d);
                  int j = RF::low(d);
                  const int i = RF::i(d) + (MODE == MODE_SELF ? j : 0);               // self: stored relative to the run's start
                  if (MODE == MODE_SELF && j >= i)
                    j += 1;
                  const uint32_t spos = (MODE == MODE_SELF) ? pB[i] : pA[i];
                  const uint32_t sc = (MODE == MODE_SELF) ? lds_c(cB,cw2,i) : lds_c(cA,cw1,i);
                  const uint32_t cpos = pB[j], cc = lds_c(cB,cw2,j);
                  const uint32_t ssign = (sc & A.sign1) != 0, csign = (cc & A.sign2) != 0;
                  const int64_t gs = (int64_t) o1;
                  const int64_t at = (gs < rem) ? O.chunk_pos + gs : nbase + (gs - rem);
#ifdef MERGE_NO_STORE
                  if (at < A.cap && spos + cpos + sc + cc + ssign + csign + plen == 0xfffffff7u)
#else
                  if (at < A.cap)
#endif
                    A.out[at] = make_seed<MODE>(plen,spos,sc & (A.sign1-1),ssign,cpos,cc & (A.sign2-1),csign);
                  o1 += 1;
                }
            }
        }
      // The other seeds, seed-parallel in windows of EWIN slots: every entry with seeds in the window leaves a descriptor at its
      // first slot there -- i | low << 8 | plen << 18 | (seeds of its run before the window) << 24 -- and a wave max-scan over
      // "slot+1 where a descriptor sits" tells every slot where its entry's run begins.  The lane of a slot then finds its
      // partner: the k-th T2 entry of the run, or, where mask bytes / strands / the entry itself drop members of the run,
      // the k-th one that stays.
      const int Tm = T - T1;
      WSYNC();                                                   // the match's key reads are done: the array becomes the window
      for (int wb = 0; wb < Tm; wb += EWIN)
        { const int wn = Tm - wb < EWIN ? Tm - wb : EWIN;        // slots of this window
          for (int x = lane; (int) (16/sizeof(typename RF::word))*x < wn; x += 64)
            ((uint4 *) ownd)[x] = make_uint4(0,0,0,0);
          WSYNC();
          { int o = off;
            #pragma unroll
            for (int r = 0; r < 4; r++)
              { const int cnt = RF::cnt(resm[r]);
                if (cnt > 0 && o + cnt > wb && o < wb + EWIN)
                  { const int before = o < wb ? wb - o : 0;
                    ownd[o + before - wb] = RF::body(resm[r]) | ((typename RF::word) (uint32_t) before << RF::CNT_SH);
                  }
                o += cnt;
              }
          }
          WSYNC();
          XPROF(12)
          int carry = 0;
          for (int s0 = 0; s0 < wn; s0 += 64)
            { const int slot = s0 + lane;                       // within the window
              int v = (slot < wn && ownd[slot] != 0) ? slot+1 : 0;         // a descriptor is never 0: plen >= 12
              v = wave_incl_scan_max_dpp(v);
              v = v > carry ? v : carry;
              carry = __builtin_amdgcn_readlane(v,63);
              if (slot < wn)
                { const int start = v-1;
                  const typename RF::word d = ownd[start];
                  const int plen = RF::plen(d);
                  int k = (slot - start) + RF::cnt(d);
                  int j = RF::low(d);
                  const int i = RF::i(d) + (MODE == MODE_SELF ? j : 0);               // self: stored relative to the run's start
                  if (plain || !RF::dirty(d))      // every member of the run counts (but the entry itself): the k-th is the k-th
                    { j += k;
                      if (MODE == MODE_SELF && j >= i)
                        j += 1;
                    }
                  else if (MODE != MODE_FLIP)
                    { // the k-th member whose mask byte is below plen (the entry itself left out in a self comparison), eight mask
                      // bytes per LDS read like the count that made k < (members that stay): the k-th one lies inside the run,
                      // whatever the bytes behind its end say
                      const uint32_t kk = (uint32_t) plen * 0x01010101u;
                      for (;;)
                        { uint64_t w;
                          __builtin_memcpy(&w,mB + j,8);
                          uint32_t bl = ((((uint32_t) w | 0x80808080u) - kk) & 0x80808080u) ^ 0x80808080u;      // 0x80 where a byte stays
                          uint32_t bh = ((((uint32_t) (w >> 32) | 0x80808080u) - kk) & 0x80808080u) ^ 0x80808080u;
                          if (MODE == MODE_SELF)
                            { const uint32_t d = (uint32_t) (i - j);
                              if (d < 4u)      bl &= ~(0x80u << (8*d));
                              else if (d < 8u) bh &= ~(0x80u << (8*(d-4)));
                            }
                          const int cl = __popc(bl), ch = __popc(bh);
                          if (k >= cl + ch)
                            { k -= cl + ch; j += 8;
                              continue;
                            }
                          uint32_t m = bl;
                          if (k >= cl) { k -= cl; m = bh; j += 4; }
                          const uint32_t m1 = m & (m-1), m2 = m1 & (m1-1), m3 = m2 & (m2-1);       // flags with the lowest 1 / 2 / 3 cleared
                          m = k == 0 ? m : (k == 1 ? m1 : (k == 2 ? m2 : m3));
                          j += (__ffs((int) m) - 1) >> 3;
                          break;
                        }
                    }
                  else
                    { const int mlen = A.soft_mask ? plen : 41;
                      for (;; j++)
                        { if (A.soft_mask && (int) mB[j] >= mlen)
                            continue;
                          if (lds_c(cB,cw2,j) & A.sign2)
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
                  const int64_t gs = (int64_t) T1 + wb + slot;
                  const int64_t at = (gs < rem) ? O.chunk_pos + gs : nbase + (gs - rem);
#ifdef MERGE_NO_STORE                   // timing experiment: everything but the seed stores (the test keeps the operands alive)
                  if (at < A.cap && spos + cpos + sc + cc + ssign + csign + plen == 0xfffffff7u)
#else
                  if (at < A.cap)
#endif
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
  XPROF(5)
  return na;
}

template <int MODE, int T2CAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(T2CAP == T2STD ? WAVE_OCC : (T2CAP > 1024 ? 1 : 2),T2CAP == T2STD ? WAVE_OCC : (T2CAP > 1024 ? 1 : 2))))
void seed_merge_walk_kernel(merge_args A)
{ __shared__ __attribute__((aligned(16))) tile_lds<T2CAP> S;
  extern __shared__ __attribute__((aligned(16))) uint8_t cdyn[];     // contig|sign words of both sides: (T2CAP+32) cw2 + (T1CAP+32) cw1 bytes

  const int lane = threadIdx.x;
  const uint32_t *idx1 = A.idx1, *idx2 = (MODE == MODE_SELF) ? A.idx1 : A.idx2;
  const int margin = A.freq + 2;
  walk_out O;
  O.chunk_pos = O.chunk_end = 0; O.tsum = 0;
#ifdef MERGE_PROF
  O.pt = clock64();
  for (int k = 0; k < 16; k++) O.pa[k] = 0;
  O.t0 = wall_clock64(); O.ntile = 0; O.nrange = 0;
#endif

  // Ranges come off eight queues, one per XCD (workgroup b runs on XCD b mod 8; range shard + 8 k is the k-th of its
  // shard): one counter word sustains only ~88 atomics per microsecond, and a wavefront's request for its next range is
  // in flight while it works on the current one.  The big ranges come first, the launch ends on the small ones.
  const int shard = (int) (blockIdx.x & 7);
  int knext = 0;
  if (lane == 0)
    knext = atomicAdd(A.next + QSTRIDE*shard,1);
  for (;;)
    { const int r = shard + 8*__builtin_amdgcn_readfirstlane(knext);
      if (r >= A.nranges)
        break;
      if (lane == 0)
        knext = atomicAdd(A.next + QSTRIDE*shard,1);
      int p = (int) A.cuts[r];
      const int pe = (int) A.cuts[r+1];
      if (p >= pe)
        continue;
#ifdef MERGE_PROF
      O.nrange += 1;
#endif
      int u = 0;                                     // index buffer in use
      // Index entries of the next XPC prefixes, two per lane (clamped at the range end).  They are fetched one tile
      // ahead and travel HBM -> LDS like the tile data, not into registers: a register result would make the compiler
      // wait for ALL outstanding vector-memory operations where the loop uses it, i.e. for the acknowledgements of the
      // seed stores the tile before has just issued.  Their arrival is covered by the tile's own wait for its data
      // (issued after them, completed in order); the reads below are opaque to the compiler for the same reason.
#define IDX_ISSUE(P0,U)                                                                                              \
      { const int last_ = pe-(P0)-1;                                                                                   \
        const int64_t e0_ = (P0) + (lane < last_ ? lane : last_), e1_ = (P0) + (lane+64 < last_ ? lane+64 : last_);    \
        G2L(idx1 + e0_,&S.ixs[U][0][4 + lane],4);                                                                      \
        G2L(idx1 + e1_,&S.ixs[U][0][68 + lane],4);                                                                     \
        if (MODE != MODE_SELF)                                                                                         \
          { G2L(idx2 + e0_,&S.ixs[U][1][4 + lane],4);                                                                  \
            G2L(idx2 + e1_,&S.ixs[U][1][68 + lane],4);                                                                 \
          }                                                                                                            \
      }
      IDX_ISSUE(p,0)
      // entries before the range: absolute (the tables may hold more than 2^32: the index arrays carry the low words, the
      // differences below are u32 arithmetic that wraps with them) -- a64 / b64 follow the low words a / b tile by tile
      int64_t a64 = fga_idx_abs(idx1,A.car1,(int64_t) p - 1), b64 = fga_idx_abs(idx2,(MODE == MODE_SELF) ? A.car1 : A.car2,(int64_t) p - 1);
      uint32_t a = (uint32_t) a64, b = (uint32_t) b64;
      VM_WAIT();
      XPROF(15)
      while (p < pe)
        { uint32_t ca0, ca1, cb0, cb1;
          { const uint32_t at = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint32_t *) &S.ixs[u][0][4 + lane];
            uint64_t va, vb;
            if (MODE == MODE_SELF)
              { asm volatile("ds_read2st64_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(va) : "v"(at) : "memory");
                vb = va;
              }
            else
              asm volatile("ds_read2st64_b32 %0, %2 offset1:1\n\tds_read2st64_b32 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                           : "=&v"(va), "=&v"(vb) : "v"(at), "v"(at + 4*(XPC+4)) : "memory");
            ca0 = (uint32_t) va; ca1 = (uint32_t) (va >> 32);
            cb0 = (uint32_t) vb; cb1 = (uint32_t) (vb >> 32);
          }
          XPROF(13)
          const int navail = pe - p < XPC ? pe - p : XPC;
          // whole panels that fit a tile
          const uint32_t cap2 = (MODE == MODE_SELF) ? T1CAP : T2CAP;
          const bool fit0 = lane < navail && (ca0 - a) <= (uint32_t) T1CAP && (cb0 - b) <= cap2;
          const bool fit1 = lane + 64 < navail && (ca1 - a) <= (uint32_t) T1CAP && (cb1 - b) <= cap2;
          int q = __popcll(__builtin_amdgcn_ballot_w64(fit0));          // the conditions are monotone in the prefix: prefix masks
          if (q == 64)
            q += __popcll(__builtin_amdgcn_ballot_w64(fit1));
          const int adv = q > 0 ? q : 1;
          uint32_t a1, b1;
          if (adv <= 64)
            { a1 = (uint32_t) __builtin_amdgcn_readlane((int) ca0,adv-1); b1 = (uint32_t) __builtin_amdgcn_readlane((int) cb0,adv-1); }
          else
            { a1 = (uint32_t) __builtin_amdgcn_readlane((int) ca1,adv-65); b1 = (uint32_t) __builtin_amdgcn_readlane((int) cb1,adv-65); }
          const int64_t n1 = (int64_t) (uint32_t) (a1 - a), n2 = (int64_t) (uint32_t) (b1 - b);
          // the entries before the tile, for the panel bounds of its keys
          if (lane == 0)
            { S.ixs[u][0][3] = a; S.ixs[u][1][3] = b; }
          // the index entries of the tile after this one are on their way while this one is processed
          const int pn = p + adv;
          XPROF(14)
          if (pn < pe)
            IDX_ISSUE(pn,u^1)
          XPROF(0)
          const uint32_t *ix1 = &S.ixs[u][0][4], *ix2 = &S.ixs[u][1][4];
          bool tiled = false;
          if (n1 > 0 && n2 > 0)
            { int lbl;
              if (q > 0)
                { walk_tile<MODE,T2CAP>(A,S,cdyn,ix1,ix2,p,true,a64,(int) n1,b64,(int) n2,0,(int) n1,false,margin,O,lbl);
                  tiled = true;
                }
              else if (MODE == MODE_SELF)
                { // one oversize panel against itself: runs of T1 entries with `margin` neighbours either side
                  const int C = (T2CAP - 2*margin) < T1CAP ? (T2CAP - 2*margin) : T1CAP;
                  for (int64_t i0 = 0; i0 < n1; i0 += C)
                    { const int64_t i1 = i0 + C < n1 ? i0 + C : n1;
                      const int64_t s0 = i0 - margin > 0 ? i0 - margin : 0, s1 = i1 + margin < n1 ? i1 + margin : n1;
                      walk_tile<MODE,T2CAP>(A,S,cdyn,ix1,ix2,p,false,a64+s0,(int) (s1-s0),a64+s0,(int) (s1-s0),(int) (i0-s0),
                                            (int) (i1-s0),false,margin,O,lbl);
                    }
                  tiled = true;
                }
              else
                { // one oversize panel, streamed: a window of the T2 panel, the T1 entries it can finish, move on
                  int64_t aw = a64, bw = b64;
                  const int64_t ae = a64 + n1, be = b64 + n2;
                  while (aw < ae)
                    { const int n1w = (int) (ae - aw < T1CAP ? ae - aw : T1CAP);
                      const int n2w = (int) (be - bw < T2CAP ? be - bw : T2CAP);
                      const bool limit = bw + n2w < be;
                      const int na = walk_tile<MODE,T2CAP>(A,S,cdyn,ix1,ix2,p,false,aw,n1w,bw,n2w,0,n1w,limit,margin,O,lbl);
                      if (na == 0)
                        { // no T1 key within reach of this window: find where the next one lands
                          const uint64_t kq = A.K1[aw];
                          int64_t l = wave_lower_bound(A.K2,bw + n2w - margin,be,kq) - margin;
                          if (l <= bw) l = bw + 1;
                          bw = l;
                          continue;
                        }
                      aw += na;
                      const int64_t nbw = bw + lbl - margin;
                      if (nbw > bw) bw = nbw;
                    }
                  tiled = true;
                }
            }
          if (!tiled)                                          // no tile, no wait of a tile: the index entries may be on their way
            VM_WAIT();
          a = a1; b = b1; a64 += n1; b64 += n2; p = pn; u ^= 1;
        }
#undef IDX_ISSUE
    }

  const merge_cold cold = *A.cold;
  if (O.chunk_end > O.chunk_pos)                       // the unused tail of the last chunk stays open: its blocks say so
    { const int64_t b0 = O.chunk_pos >> 10, b1 = (O.chunk_end - 1) >> 10;
      for (int64_t bk = b0 + lane; bk <= b1; bk += 64)
        if (bk < cold.nblocks)
          cold.valid[bk] = (uint16_t) (bk == b0 ? (O.chunk_pos & (FGA_SEED_BLOCK-1)) : 0);
      if (lane == 0)
        atomicAdd(cold.hslots,(unsigned long long) (O.chunk_end - O.chunk_pos));
    }
#ifdef MERGE_PROF
  XPROF(6)
  if (lane == 0)
    { for (int k = 0; k < 16; k++) atomicAdd(merge_prof+32*k,O.pa[k]);
      if (blockIdx.x < 8192)
        { merge_tl[4*blockIdx.x] = O.t0; merge_tl[4*blockIdx.x+1] = wall_clock64();
          merge_tl[4*blockIdx.x+2] = O.ntile; merge_tl[4*blockIdx.x+3] = O.nrange;
        }
    }
#endif
  unsigned long long tsum = O.tsum;
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0 && tsum != 0)
    atomicAdd(cold.tseed,tsum);
}


// ---------------------------------------------------------------------------------------------------
// Cutoffs beyond what the largest LDS window holds (-f > 1982; the reference takes any positive cutoff, FastGA.c:4497-4499):
// the same function of SURVEY App. B.1, evaluated straight from the field arrays in HBM.  A wavefront takes 12-mer prefixes
// in a stride; a lane owns one table-1 entry of the panel: lower bound of its key in the table-2 panel by bisection (global
// loads), plen from the two neighbours' keys, the run grown on table 2's own lcp bytes exactly as the window kernel grows it
// (first step either way decided on the keys, the following ones on the bytes, given up once more than `freq` members are
// certain), the same mask / strand / self rules, and the seeds written to a dense stretch of the buffer taken with one atomic
// per 64 entries.  No LDS, no tiles: a run of thousands of members is walked by its lane.  Two orders of magnitude slower per
// entry than the window kernel and meant for what it is -- a rarely used option on repeat-rich inputs; bit-identical seeds.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64)
void seed_merge_any_kernel(merge_args A, int pbeg, int pend)
{ const int lane = threadIdx.x;
  const uint32_t *idx1 = A.idx1, *idx2 = (MODE == MODE_SELF) ? A.idx1 : A.idx2;
  const int freq = A.freq, cw1 = A.cw1, cw2 = A.cw2;
  unsigned long long tsum = 0;
  for (int64_t p = (int64_t) pbeg + blockIdx.x; p < pend; p += gridDim.x)
    { const fga_car &cq = (MODE == MODE_SELF) ? A.car1 : A.car2;
      const int64_t a0 = fga_idx_abs(idx1,A.car1,p-1), a1 = fga_idx_abs(idx1,A.car1,p);
      const int64_t b0 = fga_idx_abs(idx2,cq,p-1), b1 = fga_idx_abs(idx2,cq,p);
      if (a1 <= a0 || b1 <= b0)
        continue;
      for (int64_t ab = a0; ab < a1; ab += 64)
        { const int64_t i = ab + lane;
          const bool act = i < a1;
          int64_t low = 0, hgh = 0;
          int plen = 0, cnt = 0;
          bool pass = false;
          if (act)
            { const uint64_t ks = A.K1[i];
              int64_t lbnd, nb, na;
              if (MODE == MODE_SELF)
                { lbnd = i; low = i; hgh = i+1; nb = i-1; na = i+1; }
              else
                { int64_t lo = b0, hi = b1;                       // lower bound of ks among the panel's keys (same prefix byte)
                  while (lo < hi)
                    { const int64_t m = (lo + hi) >> 1;
                      if (A.K2[m] < ks) lo = m+1; else hi = m;
                    }
                  lbnd = low = hgh = lo; nb = lo-1; na = lo;
                }
              const int lkb = nb >= b0 ? lcp_key(ks,A.K2[nb]) : 0, lka = na < b1 ? lcp_key(ks,A.K2[na]) : 0;
              plen = lkb > lka ? lkb : lka;
              const bool ok = plen >= 12;
              if (ok && lkb >= plen)
                { low -= 1;
                  while (low > b0 && lbnd-low <= freq && (int) A.L2[low] >= plen)
                    low -= 1;
                }
              if (ok && lka >= plen && hgh < b1 && hgh-low <= freq)
                { hgh += 1;
                  while (hgh < b1 && hgh-low <= freq && (int) A.L2[hgh] >= plen)
                    hgh += 1;
                }
              const int mlen = A.soft_mask ? plen : 41;
              pass = ok && hgh-low < freq;
              if (A.soft_mask)
                pass = pass && (int) A.M1[i] < mlen;
              if (pass)
                { if (MODE == MODE_FLIP || A.soft_mask)
                    { for (int64_t j = low; j < hgh; j++)
                        { if (A.soft_mask && (int) A.M2[j] >= mlen)
                            continue;
                          if (MODE == MODE_FLIP && (lds_c(A.C2 + (size_t) j*cw2,cw2,0) & A.sign2))
                            continue;
                          if (MODE == MODE_SELF && j == i)
                            continue;
                          cnt += 1;
                        }
                    }
                  else
                    cnt = (int) (hgh-low) - (MODE == MODE_SELF ? 1 : 0);
                }
            }
          int T;
          const int off = wave_excl_scan_add_dpp(cnt,T);
          if (T == 0)
            continue;
          unsigned long long bb = 0;
          if (lane == 0)
            bb = atomicAdd(A.count,(unsigned long long) T);
          const uint32_t blo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) bb);
          const uint32_t bhi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (bb >> 32));
          int64_t at = (int64_t) (((uint64_t) bhi << 32) | blo) + off;
          if (cnt > 0)
            { const int mlen = A.soft_mask ? plen : 41;
              const uint32_t spos = A.P1[i], sc = lds_c(A.C1 + (size_t) i*cw1,cw1,0);
              const uint32_t ssign = (sc & A.sign1) != 0;
              for (int64_t j = low; j < hgh; j++)
                { const uint32_t cc = lds_c(A.C2 + (size_t) j*cw2,cw2,0);
                  if (A.soft_mask && (int) A.M2[j] >= mlen)
                    continue;
                  if (MODE == MODE_FLIP && (cc & A.sign2))
                    continue;
                  if (MODE == MODE_SELF && j == i)
                    continue;
                  if (at < A.cap)
                    A.out[at] = make_seed<MODE>(plen,spos,sc & (A.sign1-1),ssign,A.P2[j],cc & (A.sign2-1),(cc & A.sign2) != 0);
                  at += 1;
                }
              tsum += (unsigned long long) cnt * plen;
            }
        }
    }
  #pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
    tsum += __shfl_xor(tsum,d,64);
  if (lane == 0 && tsum != 0)
    atomicAdd(A.cold->tseed,tsum);
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

// contigs on the A (which = 0) / B side of the seeds: with flip, table 1 is genome 2 and its entries become the B side
static inline int64_t mode_flip_contigs(int flip, const fga_dgix *t1, const fga_dgix *t2, int which)
{ const fga_dgix *a = flip ? t2 : t1, *b = flip ? t1 : t2;
  return which == 0 ? (int64_t) a->nctg : (int64_t) b->nctg;
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
  if (prm->freq < 1)
    { fga_set_error("fga_seed_merge: the frequency cutoff must be positive");
      return 1;
    }
  // a seed record keeps the A contig in 24 bits (beside plen) and the B contig in 30 (beside the two strand bits)
  if ((mode_flip_contigs(prm->flip,t1,t2,0) >> 24) != 0 || (mode_flip_contigs(prm->flip,t1,t2,1) >> 30) != 0)
    { fga_set_error("fga_seed_merge: more than 2^24 contigs in genome 1 (or 2^30 in genome 2) are not supported");
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
  FGA_HIP(fga_dev_enter(dev));
  const int mode = self ? MODE_SELF : (prm->flip ? MODE_FLIP : MODE_PAIR);
  // The seed buffer is taken BEFORE the forward view is made (first use as table 1 of a pair comparison): at human scale it is
  // the size of the index builder's key buffer, which lies free at this point -- taken first it fits that piece exactly,
  // taken after the view's arrays have been carved out of the piece it is a fresh 38 GB region from the driver (~1 s on a
  // device whose freed memory is still being cleared; the view's arrays are 17 GB)
  void *pre_seeds = NULL;
  if (append == NULL && capacity > 0 && mode == MODE_PAIR && t1->fview.K == NULL)
    pre_seeds = fga_dev_acquire(dev,SLOT_SEEDS,sizeof(fga_seed)*(size_t) (capacity + 2*(int64_t) dev->ncu * 32 * FGA_SEED_BLOCK));
  if (mode == MODE_PAIR && fga_dgix_make_forward(dev,(fga_dgix *) t1))
    { fga_dev_release(dev,SLOT_SEEDS,pre_seeds);
      return 1;
    }

  merge_args A;
  memset(&A,0,sizeof(A));
  const fga_view &v1 = (mode == MODE_PAIR) ? t1->fview : t1->view, &v2 = t2->view;
  A.K1 = v1.K; A.P1 = v1.P; A.C1 = (const uint8_t *) v1.C; A.M1 = v1.M; A.idx1 = v1.idx; A.cw1 = v1.cw;
  A.K2 = v2.K; A.P2 = v2.P; A.C2 = (const uint8_t *) v2.C; A.M2 = v2.M; A.idx2 = v2.idx; A.cw2 = v2.cw; A.L2 = v2.L;
  A.car1 = fga_view_car(v1); A.car2 = fga_view_car(v2);
  A.sign1 = 0x80u << (8*(t1->contbytes-1)); A.sign2 = 0x80u << (8*(t2->contbytes-1));
  A.freq = prm->freq; A.soft_mask = prm->soft_mask;
  // prefix range: (0,0) = everything; an empty range elsewhere is an empty shard (prefix cuts of a low-complexity input)
  int64_t pb = prm->prefix_begin, pe = prm->prefix_end;
  if (pb < 0) pb = 0;
  if (pe > FGA_NPREFIX) pe = FGA_NPREFIX;
  if (pb == 0 && pe <= 0) pe = FGA_NPREFIX;
  const bool empty = pb >= pe;
  const int pbeg = (int) pb, pend = (int) (empty ? pb : pe);

  // cost at both ends of the prefix range (4 tiny D2H copies)
  uint32_t c1e = 0, c2e = 0, c1b = 0, c2b = 0;
  if (!empty)
    { hipError_t ce = hipMemcpy(&c1e,v1.idx + (pend-1),4,hipMemcpyDeviceToHost);
      if (ce == hipSuccess) ce = hipMemcpy(&c2e,v2.idx + (pend-1),4,hipMemcpyDeviceToHost);
      if (ce == hipSuccess && pbeg > 0) ce = hipMemcpy(&c1b,v1.idx + (pbeg-1),4,hipMemcpyDeviceToHost);
      if (ce == hipSuccess && pbeg > 0) ce = hipMemcpy(&c2b,v2.idx + (pbeg-1),4,hipMemcpyDeviceToHost);
      if (ce != hipSuccess)
        { fga_set_error("fga_seed_merge: reading the prefix index failed: %s",hipGetErrorString(ce));
          fga_dev_release(dev,SLOT_SEEDS,pre_seeds);          // (taken before the forward view was made: not to be left behind)
          return 1;
        }
    }
  // (the index arrays hold low words: the absolute counts add the carries' high parts)
  const int64_t a1b = (int64_t) c1b + fga_idx_hi(v1,(int64_t) pbeg-1), a1e = (int64_t) c1e + fga_idx_hi(v1,(int64_t) pend-1);
  const int64_t a2b = (int64_t) c2b + fga_idx_hi(v2,(int64_t) pbeg-1), a2e = (int64_t) c2e + fga_idx_hi(v2,(int64_t) pend-1);
  const int64_t base = a1b + a2b + 2*(int64_t) pbeg;
  const int64_t total = empty ? 0 : (a1e + a2e + 2*(int64_t) pend) - base;

  fga_dseeds *S = append;
  if (S == NULL)
    { S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
      if (S == NULL)
        { fga_set_error("out of memory");
          fga_dev_release(dev,SLOT_SEEDS,pre_seeds);
          return 1;
        }
      S->dev = dev;
      if (capacity <= 0)
        capacity = (mode == MODE_PAIR ? 4 : 2)*(empty ? 0 : a1e - a1b) + (1<<20);      // two seeds per table-1 entry (the forward view holds half)
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
      S->seeds = (fga_seed *) pre_seeds;            // (assigned first: every failure below releases it through S->seeds)
      if ((err = fga_dmalloc(&counters,CTR_WORDS*sizeof(unsigned long long))) != hipSuccess ||
          (S->seeds == NULL &&
           (S->seeds = (fga_seed *) fga_dev_acquire(dev,SLOT_SEEDS,sizeof(fga_seed)*(size_t) S->phys_capacity)) == NULL) ||
          (S->valid = (uint16_t *) fga_dev_acquire(dev,SLOT_VALID,sizeof(uint16_t)*(size_t) nb)) == NULL)
        { fga_set_error("fga_seed_merge: device allocation failed: %s",hipGetErrorString(err));
          fga_pool_free(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); free(S);
          return 1;
        }
      S->slot = SLOT_SEEDS;
      S->dcount = (int64_t *) counters;
      hipMemsetAsync(counters,0,CTR_WORDS*sizeof(unsigned long long),dev->stream);
      hipMemsetD16Async((hipDeviceptr_t) S->valid,(unsigned short) FGA_SEED_BLOCK,(size_t) nb,dev->stream);
    }
  A.out = S->seeds; A.cap = phys;
  A.count = counters;
  { merge_cold cold;                                   // what the wavefronts read once, at their end: behind the counters
    cold.tseed = counters+CTR_STATS; cold.hslots = counters+CTR_STATS+1;
    cold.valid = S->valid; cold.nblocks = (S->phys_capacity + FGA_SEED_BLOCK-1) / FGA_SEED_BLOCK + 2;
    static_assert(sizeof(merge_cold) <= (CTR_WORDS-CTR_COLD)*sizeof(unsigned long long),"cold arguments live behind the counters");
    if ((err = hipMemcpyAsync(counters+CTR_COLD,&cold,sizeof(cold),hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
        (err = hipStreamSynchronize(dev->stream)) != hipSuccess)        // `cold` is on this stack frame
      { fga_set_error("fga_seed_merge: upload failed: %s",hipGetErrorString(err));
        if (append == NULL)
          { fga_pool_free(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); fga_dev_release(dev,SLOT_VALID,S->valid); free(S); }
        return 1;
      }
    A.cold = (const merge_cold *) (counters+CTR_COLD);
  }

  int rc = 1;
  int64_t hslots = 0;
  unsigned long long hc[3] = {0,0,0};
  hipEvent_t ev2 = NULL;
  void *work = NULL;
  hipEventCreate(&ev2);
  dev->last_ms[FGA_STAGE_MERGE] = dev->last_ms[FGA_STAGE_MERGE_PARTITION] = 0.f;
  if (!empty && prm->freq > FGA_MERGE_MAX_FREQ)
    { // a cutoff no LDS window holds: the kernel that reads the field arrays directly (one lane per table-1 entry)
      const int grid = dev->ncu * 16;
      hipEventRecord(dev->ev0,dev->stream);
      hipEventRecord(dev->ev1,dev->stream);
      if (mode == MODE_SELF)      hipLaunchKernelGGL(seed_merge_any_kernel<MODE_SELF>,dim3(grid),dim3(64),0,dev->stream,A,pbeg,pend);
      else if (mode == MODE_FLIP) hipLaunchKernelGGL(seed_merge_any_kernel<MODE_FLIP>,dim3(grid),dim3(64),0,dev->stream,A,pbeg,pend);
      else                        hipLaunchKernelGGL(seed_merge_any_kernel<MODE_PAIR>,dim3(grid),dim3(64),0,dev->stream,A,pbeg,pend);
      hipEventRecord(ev2,dev->stream);
    }
  else if (!empty)
    { // the sub-tile margin FREQ+2 must leave room in a window: the wide-window build takes over for large cutoffs
      // (cutoffs above 255 need the 64-bit result words: they take the 4096-entry windows whatever their margin)
      const bool wide = 2*(prm->freq + 2) > T2STD - 128, huge = prm->freq > 255;
      const int t2cap = huge ? 4096 : (wide ? 1024 : T2STD);
      const size_t dyn = (size_t) (t2cap + 32)*A.cw2 + (size_t) (T1CAP + 32)*A.cw1 + 16;
      // every wavefront of the launch is resident (they are persistent): as many as the LDS of a CU holds, at most the
      // register budget's
      const size_t lds = ((huge ? sizeof(tile_lds<4096>) : wide ? sizeof(tile_lds<1024>) : sizeof(tile_lds<T2STD>)) + dyn + 2047) / 2048 * 2048;
      int per_cu = (int) ((160*1024) / lds) - 1;       // measured: 13 x 11.6 KB are not all resident
      if (per_cu < 1) per_cu = 1;
      const int fit = per_cu;
      const int occ = huge ? 2 : (wide ? 8 : 12);      // measured: 10-12 at 100 Mbp (8: -7 %), 12-13 at 3 Gbp
      if (per_cu > occ) per_cu = occ;
      { const char *ev = getenv("FGA_MERGE_WAVES");
        if (ev != NULL && atoi(ev) > 0 && atoi(ev) <= fit && atoi(ev) <= 28) per_cu = atoi(ev);     // phys_capacity's slack covers 32 per CU
      }
      int grid = dev->ncu * per_cu;
      // ranges off the queues: two big ones per wavefront over the first five eighths of the cost, six small ones over the rest
      int nbig = grid*2, nranges = grid*8;
      { const char *e1 = getenv("FGA_MERGE_NR"), *e2 = getenv("FGA_MERGE_NBIG");          // experiments: ranges per wavefront
        if (e1 != NULL && atoi(e1) >= 2 && atoi(e1) <= 64) nranges = grid*atoi(e1);
        if (e2 != NULL && atoi(e2) >= 1 && atoi(e2) < nranges/grid) nbig = grid*atoi(e2);
      }
      if ((int64_t) nranges > total/2048 + 1)
        { nranges = (int) (total/2048) + 1; nbig = nranges/4; }
      if (nbig < 1) { nbig = 1; if (nranges < 2) nranges = 2; }
      if (grid > nranges) grid = nranges;
#ifdef MERGE_PROF
      merge_prof_grid = grid;
#endif
      // the cuts of the previous launch over the same two indices, prefix range and geometry are still good (a session
      // repeats its comparison): they stay with table 1 until its views go (FGA_MERGE_CUT_CACHE=0: cut anew every time)
      fga_dgix *own = (fga_dgix *) t1;
      static const int cache_on = getenv("FGA_MERGE_CUT_CACHE") == NULL || atoi(getenv("FGA_MERGE_CUT_CACHE")) != 0;
      const bool hit = cache_on && own->cutc.cuts != NULL && own->cutc.gen1 == v1.gen &&
                       own->cutc.gen2 == v2.gen && own->cutc.pbeg == pbeg && own->cutc.pend == pend &&
                       own->cutc.nranges == nranges && own->cutc.nbig == nbig && own->cutc.base == base && own->cutc.total == total;
      if (!hit)
        { fga_pool_free(own->cutc.cuts);
          memset(&own->cutc,0,sizeof(own->cutc));
          if (fga_dmalloc(&own->cutc.cuts,sizeof(int64_t)*(size_t) (nranges+2) + 64) != hipSuccess)
            { own->cutc.cuts = NULL;
              fga_set_error("fga_seed_merge: device allocation failed");
              goto done;
            }
        }
      int64_t *cuts = own->cutc.cuts;
      int *qhead = (int *) (counters + CTR_QUEUE);
      A.cuts = cuts; A.nranges = nranges; A.next = qhead;
      hipMemsetAsync(qhead,0,8*QSTRIDE*sizeof(int),dev->stream);
      hipEventRecord(dev->ev0,dev->stream);
      if (!hit)
        { hipLaunchKernelGGL(range_cut_kernel,dim3((nranges+1+255)/256),dim3(256),0,dev->stream,
                             A.idx1,A.idx2,A.car1,(mode == MODE_SELF) ? A.car1 : A.car2,pbeg,pend,base,total,nranges,nbig,cuts);
          own->cutc.gen1 = v1.gen; own->cutc.gen2 = v2.gen; own->cutc.pbeg = pbeg; own->cutc.pend = pend;
          own->cutc.nranges = nranges; own->cutc.nbig = nbig; own->cutc.base = base; own->cutc.total = total;
        }
      hipEventRecord(dev->ev1,dev->stream);
      if (huge)      launch_walk<4096>(mode,grid,dyn,dev->stream,A);
      else if (wide) launch_walk<1024>(mode,grid,dyn,dev->stream,A);
      else           launch_walk<T2STD>(mode,grid,dyn,dev->stream,A);
      hipEventRecord(ev2,dev->stream);
    }
  // one round trip: the three counters
  { unsigned long long *pin = (unsigned long long *) fga_dev_pinned(dev,sizeof(unsigned long long)*(CTR_STATS+2));
    if (pin == NULL)
      { fga_set_error("fga_seed_merge: pinned staging allocation failed");
        goto done;
      }
    err = hipMemcpyAsync(pin,counters,(CTR_STATS+2)*sizeof(unsigned long long),hipMemcpyDeviceToHost,dev->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(dev->stream);
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess)
      { fga_set_error("fga_seed_merge: kernel failed: %s",hipGetErrorString(err));
        goto done;
      }
    hc[0] = pin[CTR_COUNT]; hc[1] = pin[CTR_STATS]; hc[2] = pin[CTR_STATS+1];
    hslots = (int64_t) hc[2];
  }
  if (!empty)
    { hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE_PARTITION],dev->ev0,dev->ev1);
      // the merge stage = everything the launch runs: range cuts and the walk kernel
      hipEventElapsedTime(&dev->last_ms[FGA_STAGE_MERGE],dev->ev0,ev2);
    }
#ifdef MERGE_PROF
  { unsigned long long hp[16];
    static unsigned long long hpw[16*32], z[16*32];
    hipMemcpyFromSymbol(hpw,HIP_SYMBOL(merge_prof),sizeof(hpw));
    hipMemcpyToSymbol(HIP_SYMBOL(merge_prof),z,sizeof(z));
    for (int k = 0; k < 16; k++) hp[k] = hpw[32*k];
    double tot = 0; for (int k = 0; k < 16; k++) tot += (double) hp[k];
    if (tot > 0)
      fprintf(stderr,"merge phases (%% of wave cycles): tile select %.1f  load issue %.1f  load wait %.1f  window %.1f  match %.1f  emit %.1f  rest %.1f   (%.0f Mcycles)\n",
              100*hp[0]/tot,100*hp[1]/tot,100*hp[2]/tot,100*hp[3]/tot,100*hp[4]/tot,100*hp[5]/tot,100*hp[6]/tot,tot*1e-6);
    if (tot > 0)
      fprintf(stderr,"   match: bounds+search %.1f  neighbour reads %.1f  growth+count %.1f  (rest in match)   emission: scan+block %.1f  zero+descriptors %.1f  (slot loop in emit)\n",
              100*hp[8]/tot,100*hp[9]/tot,100*hp[10]/tot,100*hp[11]/tot,100*hp[12]/tot);
    if (tot > 0)
      fprintf(stderr,"   select: loop back + index read %.1f  ballots %.1f  (index issue in tile select)  range start %.1f\n",100*hp[13]/tot,100*hp[14]/tot,100*hp[15]/tot);
    static unsigned long long tl[8192*4];
    hipMemcpyFromSymbol(tl,HIP_SYMBOL(merge_tl),sizeof(tl));
    int ng = merge_prof_grid < 8192 ? merge_prof_grid : 8192;
    unsigned long long t0 = ~0ull, t1 = 0, busy = 0, tiles = 0, late = 0, tmax = 0, rmax = 0;
    for (int g = 0; g < ng; g++)
      { if (tl[4*g] < t0) t0 = tl[4*g];
        if (tl[4*g+1] > t1) t1 = tl[4*g+1];
        if (tl[4*g] > late) late = tl[4*g];
        busy += tl[4*g+1] - tl[4*g]; tiles += tl[4*g+2];
        if (tl[4*g+2] > tmax) tmax = tl[4*g+2];
        if (tl[4*g+3] > rmax) rmax = tl[4*g+3];
      }
    double span = (double) (t1 - t0);
    int hist[10] = {0,0,0,0,0,0,0,0,0,0};
    for (int g = 0; g < ng; g++)
      { int b = (int) (10.0 * (double) (tl[4*g+1] - t0) / (span + 1)); hist[b < 10 ? b : 9] += 1; }
    fprintf(stderr,"merge timeline: %d wavefronts, span %.1f us, last start at %.1f us, busy %.1f %% of span x wavefronts, tiles %llu (max %llu per wavefront), max ranges %llu\n   end-time deciles:",
            ng,span/100.0,(double) (late-t0)/100.0,100.0*(double) busy/(span*ng),tiles,tmax,rmax);
    for (int b = 0; b < 10; b++) fprintf(stderr," %d",hist[b]);
    fprintf(stderr,"\n");
  }
#endif
  S->phys_count = (int64_t) hc[0];
  S->count  = (int64_t) hc[0] - hslots;
  S->tseed  = (int64_t) hc[1];
  rc = 0;

done:
  if (ev2 != NULL) hipEventDestroy(ev2);
  fga_dev_release(dev,SLOT_TILES,work);
  if (rc != 0)
    { if (append == NULL)
        { fga_pool_free(counters); fga_dev_release(dev,SLOT_SEEDS,S->seeds); fga_dev_release(dev,SLOT_VALID,S->valid); free(S); }
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
__global__ void prefix_cut_kernel(const uint32_t *idx1, const uint32_t *idx2, fga_car car1, fga_car car2, int64_t total, int nshards, int64_t *cuts)
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
          const int64_t c = fga_idx_abs(idx1,car1,mid) + fga_idx_abs(idx2,car2,mid) + 2*((int64_t) mid+1);
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
  FGA_HIP(fga_dev_enter(dev));
  const int64_t total = t1->nents + t2->nents + 2*(int64_t) FGA_NPREFIX;
  int64_t *d = (int64_t *) fga_dev_acquire(dev,SLOT_MISC,sizeof(int64_t)*(size_t) (nshards+1));
  if (d == NULL)
    { fga_set_error("fga_merge_prefix_cuts: device allocation failed");
      return 1;
    }
  hipLaunchKernelGGL(prefix_cut_kernel,dim3((nshards+1+63)/64),dim3(64),0,dev->stream,t1->view.idx,t2->view.idx,
                     fga_view_car(t1->view),fga_view_car(t2->view),total,nshards,d);
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
