// fga_extend.hip -- O(nd) wave-based local alignment extension on MI355X (gfx950).
//
// Replaces Local_Alignment / forward_wave / reverse_wave (reference align.c:1423-1576, 352-874, 878-1418) and
// the per-hit extension loop of align_contigs (FastGA.c:3227-3341).  One wavefront owns one *unit* (a diagonal
// bucket pair of one contig pair and strand, fga_chain.c): the hits of a unit must be extended in order because
// each Local_Alignment start depends on where the previous one ended; different units are independent.
//
// Inside a wave step lane = diagonal:
//   * the three-way furthest-reaching choice reads the previous wave's V from an LDS ring (double buffered),
//     inherits the winner's 60-column match history T and trace-point list head HA,
//   * the snake compares 32 bases per step on the 2-bit packed genomes (the .bps image, unchanged, plus a
//     reverse-complemented image of genome 1 for the complement pass) with xor + ctz/clz,
//   * M = popcount(T & (2^61-1)) is an invariant of the reference's incremental bookkeeping, so it is not stored,
//   * the reference's descending-k "new best point" side effects (besta/lasta/trim*) become a wave-wide
//     exclusive prefix-max: a lane is a record iff its value beats besta and every lane before it in sweep order,
//   * sequence-end clipping and WAVE_LAG pruning are ballots,
//   * trace-point pebbles go to a per-wave arena in HBM, slots handed out by a wave-wide scan.
// The pebble chain is unwound by lane 0 (one pointer chase tip -> root), the trace is written as bytes
// (Compress_TraceTo8(ovl,0): silent truncation, align.c:3906-3908) and the accept test of FastGA.c:3264-3265 is
// evaluated in fp64.  No MFMA; latency-bound integer work, reported as such.

#include "fga_device.hpp"

#ifndef EXT_ACCOUNT
#define EXT_ACCOUNT 1                   // per-lane snake-base accounting for bench.py's `extend` block (B_ext)
#endif
#define RC          512                 // diagonals in the LDS ring
#define RMASK       (RC-1)
#define PATH_LEN    60
#define PATH_TOPB   0x1000000000000000ull
#define PATH_INT    0x0fffffffffffffffull
#define WIN61       0x1fffffffffffffffull
#define TRIM_LEN    15
#define TRIM_MASK   0x7fff
#define TRIM_MLAG   250
#define WAVE_LAG    70
#define DUB_TRIM    45
#define BIGI        0x7fffffff
#define BUCK_ANTI   128
#define TS          100                 // trace-point spacing: FastGA only ever uses TSPACE = 100 (FastGA.c:46)

#define WDW         512                 // dwords per LDS sequence window (8192 bases)
#define WINB        (WDW*16)

#define LDS_PTR __attribute__((address_space(3)))

// Trace-point ("pebble") arena of a wavefront.  A wave extension numbers its cells 0,1,2,... (the pebble pointers are
// these logical indices) and starts again at 0 with the next call.  The cells live in ONE pool shared by all wavefronts
// of the launch; a wavefront is given levels of geometrically growing size on demand -- level l holds the logical cells
// [(2^l - 1) C0, (2^(l+1) - 1) C0) -- by one atomic on the pool head, and keeps them for the rest of the launch.  So the
// pool holds what the launch really uses (a 2-kbp repeat copy needs ~500 cells, a 94-Mbp contig 2 x 10^7) instead of
// the worst case times the number of wavefronts, and the number of resident wavefronts no longer depends on the
// contig size.  (The reference grows one array per thread: enlarge_points / enlarge_vector, align.c:52-149.)
#define ARENA_L0    14                  // level 0: 16384 cells = 256 KB
#define ARENA_NLEV  16                  // levels 0..15 cover 2^30 logical cells (32-bit arena arithmetic)
#define ARENA_LEVEL(i)  (31 - __builtin_clz((((unsigned) (i)) >> ARENA_L0) + 1u))
#define ARENA_START(l)  ((int) (((1u << (l)) - 1u) << ARENA_L0))
#define ARENA_END(l)    ((int) (((2u << (l)) - 1u) << ARENA_L0))       // level 15: 2^30 - 2^14, fits an int

struct ext_seq
  { const uint32_t *img;      // padded 2-bit image as dwords (16 bases per dword, base i in bits 2*(i&15))
    int64_t base;             // base index of contig position 0 inside img
    int     len;
    LDS_PTR uint32_t *win;    // LDS window: a dword-aligned copy of img[p0/16 .. p0/16 + WDW)
    int64_t  p0;              // image base index of the window start (multiple of 16); -1: empty
    int      w0;              // contig position of the window start (p0 - base), valid when p0 >= 0
    int      bsh;             // base & 15
  };

// One workgroup = one wavefront.  LDS operations of one wavefront complete in order, so lanes see each other's
// writes without s_barrier; WAVE_SYNC() is only a compiler-level wavefront-scope fence (no re-ordering, no stale
// register copies of LDS).  A __syncthreads() here would also wait for the outstanding pebble stores to HBM.
#define WAVE_SYNC()  do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL,"wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
struct ext_shared
  { int      V[2][RC];
    int      HA[2][RC];
    int      HM[2][RC];     // mark of the pebble HA points at (saves a dependent HBM read per crossing)
    uint64_t T[2][RC];
    int      NA[RC];
    uint32_t winA[WDW+4];     // sliding windows of the two packed sequences around the wave front
    uint32_t winB[WDW+4];
    int      tsum[32];        // 5-column groups of the trim tables: sum, max prefix incl. / excl. the full group
    int      tmaxi[32];
    int      tmaxe[32];
    long long lvl_off[ARENA_NLEV];   // trace-point arena: pool cell of logical cell 0 of level l, minus the level's start
    int      nlev;                   // levels this wavefront has been given so far
  };

__shared__ ext_shared ext_lds;     // the one LDS block of a (single-wavefront) workgroup

struct ext_prof
  { unsigned long long t_steps, t_unwind, t_total, nsteps;
    unsigned long long ncells;        // diagonal updates: sum over wave steps of the wave width (cell updates)
    unsigned long long nbases;        // bases compared by the snakes (each counts once per sequence in B_ext)
  };

struct ext_state              // wave-uniform alignment state (the reference's Path + trace pointer)
  { int abpos, bbpos, aepos, bepos, diffs, tlen;
    int tpos;                 // index of trace[0] inside the per-wave trace scratch
  };

struct ext_args
  { // genomes
    const uint32_t *imgA, *imgAr, *imgB;
    const int64_t  *boffA, *boffB;       // byte offset of each contig inside the (unpadded) image
    const int64_t  *clenA, *clenB;
    const int      *permA, *permB;       // length-sorted -> original contig index
    int64_t padA, padB;                  // front padding of the images in bytes
    // work
    const fga_unit *units; const fga_hit *hits; int nunits;
    const int      *order;               // units by decreasing estimated work
    int            *next;                // work-queue head
    // alignment parameters
    int   tspace, path_ave, self, aln_min, mscore, force_lds;
    double aln_rate;
    const int16_t *table, *score;
    // scratch (per workgroup)
    int4     *pool;   int64_t pool_cells;    // the launch's cell pool (pebble levels and trace scratch of all wavefronts)
    unsigned long long *pool_next;           // its head
    // output
    fga_aln  *alns; int64_t aln_cap;
    uint8_t  *tbytes; int64_t tbytes_cap;
    unsigned long long *counters;            // [0] alignments, [1] trace bytes, [2] calls, [3] waves, [4] error flag
  };

// ---------------------------------------------------------------------------------------------------
// sequence access: 32 bases starting at contig position pos (may lie before 0 / beyond len: padding)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t fetch32(const ext_seq &s, int64_t pos)
{ int64_t p = s.base + pos;
  int sh = (int) (p & 15) * 2;
  uint32_t d0, d1, d2;
  int64_t q = p - s.p0;
  if (s.p0 >= 0 && q >= 0 && q + 48 <= WINB)          // inside the LDS window
    { int w = (int) (q >> 4);
      d0 = s.win[w]; d1 = s.win[w+1]; d2 = s.win[w+2];
    }
  else
    { int64_t w = p >> 4;
      d0 = s.img[w]; d1 = s.img[w+1]; d2 = s.img[w+2];
    }
  uint64_t lo = ((uint64_t) d1 << 32) | d0;
  uint64_t v = lo >> sh;
  if (sh)
    v |= (uint64_t) d2 << (64-sh);
  return v;
}

// 64 bases starting at contig position pos as two 64-bit words (lo = first 32 bases); all window arithmetic
// is 32-bit and relative to the window start
__device__ __forceinline__ void fetch64(const ext_seq &s, int pos, uint64_t &lo, uint64_t &hi)
{ const int sh = ((s.bsh + pos) & 15) * 2;
  uint32_t d0, d1, d2, d3, d4;
  const int q = pos - s.w0;
  if (s.p0 >= 0 && q >= 0 && q + 80 <= WINB)
    { const int w = q >> 4;
      d0 = s.win[w]; d1 = s.win[w+1]; d2 = s.win[w+2]; d3 = s.win[w+3]; d4 = s.win[w+4];
    }
  else
    { const int64_t w = (s.base + pos) >> 4;
      d0 = s.img[w]; d1 = s.img[w+1]; d2 = s.img[w+2]; d3 = s.img[w+3]; d4 = s.img[w+4];
    }
  const uint64_t a = ((uint64_t) d1 << 32) | d0, b = ((uint64_t) d3 << 32) | d2;
  if (sh)
    { lo = (a >> sh) | (b << (64-sh));
      hi = (b >> sh) | ((uint64_t) d4 << (64-sh));
    }
  else
    { lo = a; hi = b; }
}

// number of equal bases going forward from (ax,bx), at most lim; 64 bases per step
__device__ __forceinline__ int match_fwd(const ext_seq &A, const ext_seq &B, int ax, int bx, int lim)
{ int L = 0;
  while (L < lim)
    { uint64_t alo, ahi, blo, bhi;
      fetch64(A,ax+L,alo,ahi);
      fetch64(B,bx+L,blo,bhi);
      uint64_t x = alo ^ blo, y = ahi ^ bhi;
      if (x != 0)
        { L += (__ffsll((unsigned long long) x) - 1) >> 1;
          break;
        }
      if (y != 0)
        { L += 32 + ((__ffsll((unsigned long long) y) - 1) >> 1);
          break;
        }
      L += 64;
    }
  return L < lim ? L : lim;
}

// number of equal bases going backward: A[ax-1]==B[bx-1], A[ax-2]==B[bx-2], ... at most lim
__device__ __forceinline__ int match_rev(const ext_seq &A, const ext_seq &B, int ax, int bx, int lim)
{ int L = 0;
  while (L < lim)
    { uint64_t alo, ahi, blo, bhi;
      fetch64(A,ax-L-64,alo,ahi);
      fetch64(B,bx-L-64,blo,bhi);
      uint64_t x = alo ^ blo, y = ahi ^ bhi;      // y holds the 32 bases nearest to (ax,bx)
      if (y != 0)
        { L += __clzll((long long) y) >> 1;
          break;
        }
      if (x != 0)
        { L += 32 + (__clzll((long long) x) >> 1);
          break;
        }
      L += 64;
    }
  return L < lim ? L : lim;
}

struct wseq;
__device__ __forceinline__ int base_at(const wseq &s, int pos);

// ---------------------------------------------------------------------------------------------------
// wave-wide helpers (all 64 lanes must call)
// ---------------------------------------------------------------------------------------------------
// Wave64 scans on the DPP cross-lane network (row_shr 1/2/4/8 inside rows of 16, then row_bcast:15 / row_bcast:31
// across rows) instead of ds_bpermute round trips through the LDS crossbar: 6 VALU-rate ops per scan.
#define DPP_STEP(OPX,CTRL,RMASK_)                                                   \
  { int _t = __builtin_amdgcn_update_dpp(ID,x,CTRL,RMASK_,0xf,false); x = OPX; }

__device__ __forceinline__ int rdlane(int v, int l)      // l is wave-uniform
{ return __builtin_amdgcn_readlane(v,l); }

__device__ __forceinline__ int wscan_add_excl(int v, int &total)
{ const int ID = 0;
  int x = v;
  DPP_STEP(x+_t,0x111,0xf) DPP_STEP(x+_t,0x112,0xf) DPP_STEP(x+_t,0x114,0xf) DPP_STEP(x+_t,0x118,0xf)
  DPP_STEP(x+_t,0x142,0xa) DPP_STEP(x+_t,0x143,0xc)
  total = rdlane(x,63);
  return x - v;
}

template <int S>
__device__ __forceinline__ int wscan_best_excl(int v)     // exclusive prefix max (S>0) / min (S<0) in lane order
{ const int ID = (S > 0) ? -BIGI : BIGI;
  int x = v;
  if (S > 0)
    { DPP_STEP(x > _t ? x : _t,0x111,0xf) DPP_STEP(x > _t ? x : _t,0x112,0xf) DPP_STEP(x > _t ? x : _t,0x114,0xf)
      DPP_STEP(x > _t ? x : _t,0x118,0xf) DPP_STEP(x > _t ? x : _t,0x142,0xa) DPP_STEP(x > _t ? x : _t,0x143,0xc)
    }
  else
    { DPP_STEP(x < _t ? x : _t,0x111,0xf) DPP_STEP(x < _t ? x : _t,0x112,0xf) DPP_STEP(x < _t ? x : _t,0x114,0xf)
      DPP_STEP(x < _t ? x : _t,0x118,0xf) DPP_STEP(x < _t ? x : _t,0x142,0xa) DPP_STEP(x < _t ? x : _t,0x143,0xc)
    }
  return __builtin_amdgcn_update_dpp(ID,x,0x138,0xf,0xf,false);     // wave_shr:1 -> exclusive
}

__device__ __forceinline__ int last_lane(uint64_t m)  { return 63 - __clzll((long long) m); }
__device__ __forceinline__ int first_lane(uint64_t m) { return __ffsll((unsigned long long) m) - 1; }

// TABLE / SCORE of the reference's Align_Spec (align.c:207-218) without the two 64 KB tables: for a 15-bit match
// pattern p (most significant bit = oldest column), match = +ms, mismatch = -(1000-ms),
//   score(p) = sum over the 15 columns,   table(p) = score(p) - max over proper prefixes (incl. empty) of the prefix sum,
// both truncated to int16 like the stored tables.  The pattern is cut into three 5-column groups whose
// (sum, max prefix) come from a 32-entry LDS table filled at kernel start: 3 independent LDS reads + ~10 integer ops.
__device__ __forceinline__ int trim_score(uint32_t p, int ms)
{ int ones = __popc(p);
  return (int) (int16_t) (ms*ones - (1000-ms)*(15-ones));
}

__device__ __forceinline__ void trim_fill(LDS_PTR ext_shared *sh, int ms)
{ const int lane = threadIdx.x & 63;
  if (lane < 32)
    { int score = 0, mi = 0, me = 0;
      const int ds = 1000-ms;
      #pragma unroll
      for (int i = 4; i >= 0; i--)
        { score += ((lane >> i) & 1) ? ms : -ds;
          if (i > 0) { me = score > me ? score : me; }
          mi = score > mi ? score : mi;
        }
      sh->tsum[lane] = score; sh->tmaxi[lane] = mi; sh->tmaxe[lane] = me;
    }
}

__device__ __forceinline__ int trim_table(LDS_PTR ext_shared *sh, uint32_t p)
{ const uint32_t g1 = (p >> 10) & 31, g2 = (p >> 5) & 31, g3 = p & 31;
  const int s1 = sh->tsum[g1], s2 = sh->tsum[g2], s3 = sh->tsum[g3];
  const int m1 = sh->tmaxi[g1], m2 = s1 + sh->tmaxi[g2], m3 = s1 + s2 + sh->tmaxe[g3];
  int mx = m1 > m2 ? m1 : m2;
  mx = mx > m3 ? mx : m3;
  return (int) (int16_t) (s1 + s2 + s3 - mx);
}

// (re)load the LDS window of a sequence so that contig position `pos` sits `before` bases after its start;
// wave-uniform, all lanes participate; the caller synchronises before the next fetch
__device__ __forceinline__ void win_load(ext_seq &s, int pos, int before)
{ int64_t p0 = (s.base + pos - before) & ~(int64_t) 15;
  if (p0 < 0) p0 = 0;
  const uint32_t *g = s.img + (p0 >> 4);
  for (int i = threadIdx.x & 63; i < WDW; i += 64)
    s.win[i] = g[i];
  s.p0 = p0;
  s.w0 = (int) (p0 - s.base);
}

// keep [pos-lo, pos+hi] inside the window; S > 0 keeps most of the window ahead of pos, S < 0 behind it
template <int S>
__device__ __forceinline__ bool win_track(ext_seq &s, int pos)
{ const int q = pos - s.w0;
  if (S > 0)
    { if (s.p0 >= 0 && q >= 448 && q + 1536 <= WINB)
        return false;
      win_load(s,pos,512);
    }
  else
    { if (s.p0 >= 0 && q + 448 <= WINB && q >= 1536)
        return false;
      win_load(s,pos,WINB-512);
    }
  return true;
}

// ---------------------------------------------------------------------------------------------------
// wave-uniform values.  Arguments of a non-inlined device function arrive in VGPRs and everything derived from
// them is treated as divergent (exec-mask branches, VALU bookkeeping); readfirstlane moves them to SGPRs so the
// wave bookkeeping (low/hgh/besta/..., window tests, branches) runs on the scalar unit.
// ---------------------------------------------------------------------------------------------------
#define GLB_PTR __attribute__((address_space(1)))
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));      // a pebble cell as a plain 16-byte vector (int4 layout)
#define UNI(v) __builtin_amdgcn_readfirstlane((int) (v))
#define LIKELY(c)   __builtin_expect(!!(c),1)      // block placement: the common path of a wave step falls through,
#define UNLIKELY(c) __builtin_expect(!!(c),0)      // a taken branch costs a single wavefront an instruction refetch
#define BALLOT(p) __builtin_amdgcn_ballot_w64(p)
__device__ __forceinline__ int64_t uni64(int64_t v)
{ const uint32_t lo = (uint32_t) UNI((uint32_t) v), hi = (uint32_t) UNI((uint32_t) ((uint64_t) v >> 32));
  return (int64_t) (((uint64_t) hi << 32) | lo);
}

struct wseq                   // ext_seq with every field wave-uniform and explicit address spaces
  { const GLB_PTR uint32_t *img;
    LDS_PTR uint32_t *win;
    int64_t base;
    int     len, w0, bsh;     // window start as a contig position; base & 15
  };

__device__ __forceinline__ void wseq_from(wseq &W, const ext_seq &s, LDS_PTR uint32_t *win)
{ W.img  = (const GLB_PTR uint32_t *) uni64((int64_t) s.img);
  W.win  = win;
  W.base = uni64(s.base);
  W.len  = UNI(s.len);
  W.bsh  = (int) (W.base & 15);
  W.w0   = UNI(s.w0);
}

__device__ __forceinline__ void wseq_load(wseq &s, int pos, int before)     // see win_load
{ int64_t p0 = (s.base + pos - before) & ~(int64_t) 15;
  if (p0 < 0) p0 = 0;
  const GLB_PTR uint32_t *g = s.img + (p0 >> 4);
  for (int i = threadIdx.x & 63; i < WDW; i += 64)
    s.win[i] = g[i];
  s.w0 = (int) (p0 - s.base);
}

__device__ __forceinline__ int base_at(const wseq &s, int pos)     // 0..3, or 4 outside [0,len)
{ if (pos < 0 || pos >= s.len)
    return 4;
  int64_t p = s.base + pos;
  return (s.img[p >> 4] >> ((p & 15)*2)) & 3;
}

template <int S>
__device__ __forceinline__ void wseq_track(wseq &s, int pos)
{ const int q = pos - s.w0;
  if (S > 0)
    { if (q < 448 || q + 1536 > WINB)
        wseq_load(s,pos,512);
    }
  else
    { if (q + 448 > WINB || q < 1536)
        wseq_load(s,pos,WINB-512);
    }
}

// One round of a snake: the number of equal bases (0..64) from (pa,pb) in direction S, where pa/pb is the lowest
// position of the 64-base stretch.  Ten LDS dwords issued together, aligned with one v_alignbit per dword.
template <int S>
__device__ __forceinline__ int snake_cmp(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t sa,
                                         uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t sb)
{ const uint32_t x0 = __builtin_amdgcn_alignbit(a1,a0,sa) ^ __builtin_amdgcn_alignbit(b1,b0,sb);
  const uint32_t x1 = __builtin_amdgcn_alignbit(a2,a1,sa) ^ __builtin_amdgcn_alignbit(b2,b1,sb);
  const uint32_t x2 = __builtin_amdgcn_alignbit(a3,a2,sa) ^ __builtin_amdgcn_alignbit(b3,b2,sb);
  const uint32_t x3 = __builtin_amdgcn_alignbit(a4,a3,sa) ^ __builtin_amdgcn_alignbit(b4,b3,sb);
  // position of the first difference as a bit index into x3:x2:x1:x0 (forward) / from the top (backward), 128 if none;
  // v_ffbl / v_ffbh return -1 for a zero word, which the unsigned min chain turns into "look further"
  uint32_t f;
  if (S > 0)
    { uint32_t f0 = x0 ? (uint32_t) __builtin_ctz(x0) : 128u, f1 = x1 ? 32u + (uint32_t) __builtin_ctz(x1) : 128u;
      uint32_t f2 = x2 ? 64u + (uint32_t) __builtin_ctz(x2) : 128u, f3 = x3 ? 96u + (uint32_t) __builtin_ctz(x3) : 128u;
      f0 = f0 < f1 ? f0 : f1; f2 = f2 < f3 ? f2 : f3;
      f = f0 < f2 ? f0 : f2;
    }
  else          // x3 holds the 16 bases nearest to (ax,bx)
    { uint32_t f3 = x3 ? (uint32_t) __builtin_clz(x3) : 128u, f2 = x2 ? 32u + (uint32_t) __builtin_clz(x2) : 128u;
      uint32_t f1 = x1 ? 64u + (uint32_t) __builtin_clz(x1) : 128u, f0 = x0 ? 96u + (uint32_t) __builtin_clz(x0) : 128u;
      f3 = f3 < f2 ? f3 : f2; f1 = f1 < f0 ? f1 : f0;
      f = f3 < f1 ? f3 : f1;
    }
  return (int) (f >> 1);
}

template <int S>
__device__ __attribute__((noinline)) int snake_hbm(const GLB_PTR uint32_t *ga, const GLB_PTR uint32_t *gb, uint32_t sa, uint32_t sb)
{ // rare: a stretch outside the LDS windows, straight from the HBM images.  Kept out of line so that the compiler's
  // waitcnt bookkeeping of the common path never sees an outstanding global load (it would make every wave step
  // wait for the acknowledgement of the previous pebble stores).
  const uint32_t a0 = ga[0], a1 = ga[1], a2 = ga[2], a3 = ga[3], a4 = ga[4];
  const uint32_t b0 = gb[0], b1 = gb[1], b2 = gb[2], b3 = gb[3], b4 = gb[4];
  return snake_cmp<S>(a0,a1,a2,a3,a4,sa,b0,b1,b2,b3,b4,sb);
}

template <int S>
__device__ __forceinline__ int snake_round(const wseq &A, const wseq &B, int pa, int pb)
{ const int qa = pa - A.w0, qb = pb - B.w0;
  const uint32_t sa = (uint32_t) ((A.bsh + pa) & 15) * 2, sb = (uint32_t) ((B.bsh + pb) & 15) * 2;
  const bool inw = (uint32_t) qa <= (uint32_t) (WINB-80) && (uint32_t) qb <= (uint32_t) (WINB-80);
  int n;
  if (__builtin_expect(BALLOT(!inw) == 0,1))
    { LDS_PTR const uint32_t *wa = A.win + (qa >> 4), *wb = B.win + (qb >> 4);
      n = snake_cmp<S>(wa[0],wa[1],wa[2],wa[3],wa[4],sa,wb[0],wb[1],wb[2],wb[3],wb[4],sb);
    }
  else
    n = snake_hbm<S>(A.img + ((A.base + pa) >> 4),B.img + ((B.base + pb) >> 4),sa,sb);
  return n;
}

// length of the snake from (ax,bx): equal bases going forward (S > 0: A[ax+i] == B[bx+i]) or backward (S < 0:
// A[ax-1-i] == B[bx-1-i]), at most lim.  The first 64-base round is straight-line code (reading past lim only
// touches padding or the neighbouring contig, and the result is clamped); longer snakes loop.
template <int S>
__device__ __forceinline__ int snake(const wseq &A, const wseq &B, int ax, int bx, int lim)
{ int n = snake_round<S>(A,B,(S > 0) ? ax : ax-64,(S > 0) ? bx : bx-64);
  int L = n;
  if (__builtin_expect(BALLOT(n == 64 && lim > 64) != 0,0))
    while (n == 64 && L < lim)
      { n = snake_round<S>(A,B,(S > 0) ? ax+L : ax-L-64,(S > 0) ? bx+L : bx-L-64);
        L += n;
      }
  return L < lim ? L : lim;
}

// exclusive prefix max in lane order of non-negative values (0 for lane 0): zero is the identity, so every step
// is one fused v_max_i32_dpp
__device__ __forceinline__ int wscan_max_excl_nn(int v)
{ int x = v, t;
  t = __builtin_amdgcn_update_dpp(0,x,0x111,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x112,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x114,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x118,0xf,0xf,true); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x142,0xa,0xf,false); x = x > t ? x : t;
  t = __builtin_amdgcn_update_dpp(0,x,0x143,0xc,0xf,false); x = x > t ? x : t;
  return __builtin_amdgcn_update_dpp(0,x,0x138,0xf,0xf,false);
}

// both trim-table tests of a "good" point (align.c:737-741) with their six LDS reads in one round trip
__device__ __forceinline__ bool trim_ok(LDS_PTR ext_shared *sh, uint64_t b, int ms)
{ const uint32_t qlo = (uint32_t) b & TRIM_MASK, qhi = (uint32_t) (b >> TRIM_LEN) & TRIM_MASK;
  const int tlo = trim_table(sh,qlo), thi = trim_table(sh,qhi);
  return (tlo >= 0) & (thi + trim_score(qlo,ms) >= 0);
}

// ---------------------------------------------------------------------------------------------------
// unwind the pebble chain of one wave extension into trace pairs (align.c:805-870 / 1325-1415); lane 0 chases
// the pointers, the others wait.  Shared by the register and the LDS-ring wave routines.
// ---------------------------------------------------------------------------------------------------
// The chase is a chain of dependent reads (an L2 round trip each); but a pebble's predecessor was created only a few
// wave steps earlier, i.e. a few cells lower in the arena.  So the wavefront loads a coalesced window of 64
// consecutive cells (lane l holds cells[base+l]) and follows the chain inside it with v_readlane -- about twenty
// links per memory round trip instead of one.
// take `ncell` cells off the pool (wave-uniform; lane 0 does the atomic): first cell, or -1 when the pool is exhausted
__device__ __forceinline__ long long pool_take(const ext_args &G, long long ncell)
{ unsigned long long b = 0;
  if ((threadIdx.x & 63) == 0)
    b = atomicAdd(G.pool_next,(unsigned long long) ncell);
  const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) b);
  const uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (b >> 32));
  const long long at = (long long) (((unsigned long long) hi << 32) | lo);
  return (at + ncell <= G.pool_cells) ? at : -1;
}

// make sure levels 0..need exist for this wavefront; false: pool exhausted (or level beyond the 32-bit arena)
__device__ __attribute__((noinline)) bool arena_ensure(const ext_args &G, int need)
{ LDS_PTR ext_shared *sh = (LDS_PTR ext_shared *) &ext_lds;
  int have = __builtin_amdgcn_readfirstlane(sh->nlev);
  if (need >= ARENA_NLEV)
    return false;
  while (have <= need)
    { const long long at = pool_take(G,(long long) (ARENA_END(have) - ARENA_START(have)));
      if (at < 0)
        return false;
      if ((threadIdx.x & 63) == 0)
        { sh->lvl_off[have] = at - ARENA_START(have);
          sh->nlev = have+1;
        }
      have += 1;
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL,"wavefront"); __builtin_amdgcn_wave_barrier();
  return true;
}

// the cell with logical index i (any level this wavefront holds); one LDS read for the level's offset
#define ARENA_CELL(pool,i)  ((pool) + (((LDS_PTR ext_shared *) &ext_lds)->lvl_off[ARENA_LEVEL(i)] + (long long) (i)))

struct cell_window
  { const GLB_PTR v4i *pool;
    int base, lim;            // window = logical cells [base, base+lim), lim <= 64, all inside one level
    v4i w;
  };

template <int DIR>             // DIR < 0: the walk goes to lower indices, DIR > 0: to higher ones
__device__ __forceinline__ v4i cw_get(cell_window &W, int idx)      // idx is wave-uniform
{ if (idx < W.base || idx >= W.base + W.lim)
    { // a window never straddles two levels (they are different pool stretches): it is clamped at the level's start
      // when walking down and cut at the level's end when walking up (the lanes beyond read padding / foreign cells)
      const int lv = ARENA_LEVEL(idx), ls = ARENA_START(lv), le = ARENA_END(lv);
      W.base = (DIR < 0) ? (idx-63 > ls ? idx-63 : ls) : idx;
      W.lim  = le - W.base < 64 ? le - W.base : 64;
      const long long off = ((LDS_PTR ext_shared *) &ext_lds)->lvl_off[lv];
      W.w = W.pool[off + (long long) W.base + (int) (threadIdx.x & 63)];
    }
  const int l = idx - W.base;
  v4i r;
  r.x = rdlane(W.w.x,l); r.y = rdlane(W.w.y,l); r.z = rdlane(W.w.z,l); r.w = rdlane(W.w.w,l);
  return r;
}

// Trace pairs (and reversed pebble pointers) are not stored one by one -- a global store per link makes every link
// wait for the previous store's acknowledgement as soon as the next window is read -- but collected in
// one VGPR, lane j = j-th value of the batch, and written 64 at a time.
struct lane_batch
  { uint32_t val;       // lane j: j-th value pushed since the last flush
    uint32_t key;       // lane j: its destination (scattered batches only)
    int      n;         // wave-uniform fill
  };
__device__ __forceinline__ void lb_push(lane_batch &B, uint32_t v, uint32_t k)     // v, k wave-uniform
{ const bool mine = (int) (threadIdx.x & 63) == B.n;
  B.val = mine ? v : B.val;
  B.key = mine ? k : B.key;
  B.n += 1;
}
// contiguous, descending: the j-th value goes to dword (top - j) of d32
__device__ __forceinline__ void lb_flush_desc(lane_batch &B, GLB_PTR uint32_t *d32, int64_t top)
{ const int lane = threadIdx.x & 63;
  if (lane < B.n)
    d32[top - lane] = B.val;
  B.n = 0;
}
// scattered: the j-th value goes to the first dword of cell key_j
__device__ __forceinline__ void lb_flush_cells(lane_batch &B, GLB_PTR v4i *pool)
{ const int lane = threadIdx.x & 63;
  if (lane < B.n)
    ((GLB_PTR int *) ARENA_CELL(pool,B.key))[0] = (int) B.val;
  B.n = 0;
}

template <int S>
__device__ __attribute__((noinline)) void ext_unwind(const ext_args &G, uint16_t *trace_in, int tcap_in, ext_state &P,
                                                     ext_prof &PF, int mida_in, int aoff_in, int trima_in, int trimx_in,
                                                     int trimd_in, int trimha_in, int &mind)
{ const int lane = threadIdx.x & 63;
  const int ts = TS;
  const int mida = UNI(mida_in), aoff = UNI(aoff_in), trima = UNI(trima_in), trimx = UNI(trimx_in);
  const int trimd = UNI(trimd_in), trimha = UNI(trimha_in);
  GLB_PTR v4i *cells = (GLB_PTR v4i *) uni64((int64_t) G.pool);
  GLB_PTR uint16_t *trace = (GLB_PTR uint16_t *) uni64((int64_t) trace_in);
  const bool l0 = (lane == 0);
  // the reference picks the "more" tip only when spec->reach is set; FastGA always passes reach = 0 (FastGA.c:3757)
  const int trimy = trima - trimx;
  int tlen = UNI(P.tlen), tpos = UNI(P.tpos);
  int rootk = 0;
  const unsigned long long tun = clock64();
  __syncthreads();          // once per call: all pebble stores of the wave are complete before the pointer chase
  cell_window W;
  W.pool = cells; W.base = -1000; W.lim = 0; W.w = (v4i) { 0,0,0,0 };

  if (S > 0)
    { // single walk tip -> root; the pairs come out last-to-first and are stored downwards from the top of the
      // scratch, so no count pass is needed (the reverse wave prepends below tpos afterwards)
      int pos = UNI(tcap_in) - 8;
      const int tend = pos;
      GLB_PTR uint32_t *tr32 = (GLB_PTR uint32_t *) trace;      // trace is dword aligned and every pos is even
      lane_batch LB; LB.val = 0; LB.key = 0; LB.n = 0;
      int ptop = 0;                                             // dword index of the batch's first pair
      v4i cur4 = cw_get<-1>(W,trimha);
      int lastb = 0, lastd = 0, lastk = 0;
      if (cur4.x >= 0)
        { lastb = cur4.w - cur4.y; lastd = cur4.z; lastk = cur4.y; }
      while (cur4.x >= 0)
        { const v4i prv = cw_get<-1>(W,cur4.x);
          const int bcur = cur4.w - cur4.y;
          const int bprv = (prv.x >= 0) ? prv.w - prv.y : ((mida - prv.y) >> 1);
          const int dprv = (prv.x >= 0) ? prv.z : 0;
          pos -= 2;
          if (LB.n == 0)
            ptop = pos >> 1;
          lb_push(LB,((uint32_t) (cur4.z - dprv) & 0xffffu) | ((uint32_t) (bcur - bprv) << 16),0u);
          if (LB.n == 64)
            lb_flush_desc(LB,tr32,ptop);
          cur4 = prv;
        }
      lb_flush_desc(LB,tr32,ptop);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST,"workgroup");
      rootk = cur4.y;
      if (pos == tend)
        { lastb = (mida - rootk) >> 1; lastd = 0; lastk = rootk; }
      int atlen = tend - pos;
      GLB_PTR uint16_t *at = trace + pos;
      if (lastb + lastk != trimx)
        { if (l0)
            { at[atlen]   = (uint16_t) (trimd - lastd);
              at[atlen+1] = (uint16_t) (trimy - lastb);
            }
          atlen += 2;
        }
      else if (lastb != trimy)
        { if (l0)
            { at[atlen-1] = (uint16_t) (at[atlen-1] + (trimy - lastb));
              at[atlen-2] = (uint16_t) (at[atlen-2] + (trimd - lastd));
            }
        }
      tlen = atlen;
      tpos = pos;
    }
  else
    { // reverse wave: the list runs tip (towards the alignment start) -> root (at the mid point); the reference
      // reverses it and walks root -> tip, prepending pairs.  Pass 1 reverses the pointers in place exactly like the
      // reference (align.c:1342-1348); pass 2 walks the reversed list through windows that extend upwards.
      int a = -1, h = trimha;
      lane_batch LB; LB.val = 0; LB.key = 0; LB.n = 0;
      while (h >= 0)
        { const int bq = cw_get<-1>(W,h).x;
          lb_push(LB,(uint32_t) a,(uint32_t) h);
          if (LB.n == 64)
            lb_flush_cells(LB,cells);
          a = h;
          h = bq;
        }
      lb_flush_cells(LB,cells);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST,"workgroup");       // the reversed pointers are visible to pass 2
      __builtin_amdgcn_s_waitcnt(0);
      W.base = -1000; W.lim = 0;
      h = a;
      GLB_PTR uint16_t *at = trace + tpos;
      int atlen = 0;
      v4i c4 = cw_get<+1>(W,h);
      int k = c4.y;
      int b = c4.w - k, e = 0, d = 0, aa = 0;
      if ((b+k) % ts != aoff)
        { h = c4.x;
          if (h < 0)
            { aa = trimy; d = trimd; }
          else
            { c4 = cw_get<+1>(W,h);
              k = c4.y; aa = c4.w - k; d = c4.z;
            }
          if (tlen == 0)
            { atlen -= 2;
              if (l0)
                { at[atlen+1] = (uint16_t) (b-aa);
                  at[atlen]   = (uint16_t) (d-e);
                }
            }
          else if (l0)
            { at[1] = (uint16_t) (at[1] + (b-aa));
              at[0] = (uint16_t) (at[0] + (d-e));
            }
          b = aa;
          e = d;
        }
      if (h >= 0)
        { GLB_PTR uint32_t *at32 = (GLB_PTR uint32_t *) at;    // tpos is even
          int ptop = 0;
          for (h = c4.x; h >= 0; h = c4.x)
            { c4 = cw_get<+1>(W,h);
              k = c4.y;
              aa = c4.w - k;
              d = c4.z;
              atlen -= 2;
              if (LB.n == 0)
                ptop = atlen >> 1;
              lb_push(LB,((uint32_t) (d-e) & 0xffffu) | ((uint32_t) (b-aa) << 16),0u);
              if (LB.n == 64)
                lb_flush_desc(LB,at32,ptop);
              b = aa;
              e = d;
            }
          lb_flush_desc(LB,at32,ptop);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST,"workgroup");
          if (b+k != trimx)
            { atlen -= 2;
              if (l0)
                { at[atlen+1] = (uint16_t) (b-trimy);
                  at[atlen]   = (uint16_t) (trimd-e);
                }
            }
          else if (b != trimy)
            { if (l0)
                { at[atlen+1] = (uint16_t) (at[atlen+1] + (b-trimy));
                  at[atlen]   = (uint16_t) (at[atlen]   + (trimd-e));
                }
            }
        }
      tlen = tlen - atlen;
      tpos = tpos + atlen;
    }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST,"workgroup");           // lane 0's trace pairs before any lane reads them
  P.tlen = tlen;
  P.tpos = tpos;
  if (S > 0)
    { P.aepos = trimx; P.bepos = trimy; P.diffs = trimd;
      mind = rootk;
    }
  else
    { P.abpos = trimx; P.bbpos = trimy; P.diffs = P.diffs + trimd; }
  PF.t_unwind += clock64() - tun;
}

// ---------------------------------------------------------------------------------------------------
// One directional wave extension (S = +1 forward_wave align.c:352-874, S = -1 reverse_wave align.c:878-1418).
// returns 0 ok, 1 pebble arena full, 2 wave wider than the LDS ring.
//
// Two representations of the per-diagonal state (V, T, HA, HM, NA), switched on the fly by the wave width:
//   REGISTER mode (width <= 60, the common case): lane l holds diagonal k = kref - S*l, so lane order IS the
//     reference's sweep order for both directions; V[k-S] sits in lane l+1 and V[k+S] in lane l-1 and travel over
//     DPP wave_shl:1 / wave_shr:1 -- no LDS traffic except the two sequence windows.  When the wave drifts to
//     the edge of the 64 lanes every register is rotated (ds_bpermute, rare).
//   RING mode (wider waves): the state lives in a double-buffered LDS ring indexed by k & 511 and the wave is
//     swept in chunks of 64 diagonals.
//   A wave that grows beyond 60 diagonals spills its registers to the ring and continues there; when it has
//     shrunk to <= 40 it is reloaded into registers.
// ---------------------------------------------------------------------------------------------------
// lane l gets lane l+1 / l-1.  The active lanes are kept inside [1,62] and lanes 0 and 63 hold VNEW, so what the
// two end lanes receive is never used: no `old` operand to initialise.
#define FROM_NEXT(v,oldv) __builtin_amdgcn_mov_dpp(v,0x130,0xf,0xf,true)
#define FROM_PREV(v,oldv) __builtin_amdgcn_mov_dpp(v,0x138,0xf,0xf,true)
#define REG_MAXW 60
#define REG_BACK 40
#define KOF(l)    ((S > 0) ? kref - (l) : kref + (l))
#define LOF(kk)   ((S > 0) ? kref - (kk) : (kk) - kref)
#define WIN_BACK() { Ain.p0 = A.base + A.w0; Ain.w0 = A.w0; Bin.p0 = B.base + B.w0; Bin.w0 = B.w0; }
#define BAIL(code) { WIN_BACK() return code; }

template <int S>
__device__ __attribute__((noinline)) int ext_wave(const ext_args &G, LDS_PTR ext_shared *shp_in, uint16_t *trace, int tcap,
                        ext_seq &Ain, ext_seq &Bin, ext_state &P,
                        int &mind, int maxd_in, int mida_in, int minp_in, int maxp_in, int aoff_in,
                        unsigned long long &nwaves_out, ext_prof &PF)
{ const int lane = threadIdx.x & 63;
  const unsigned long long tstart = clock64();
  // everything the wave loop touches lives in registers (by-reference arguments of a non-inlined device function
  // sit in scratch memory), and everything wave-uniform is moved to SGPRs first
  LDS_PTR ext_shared *shp = (LDS_PTR ext_shared *) &ext_lds;
  wseq A, B;
  wseq_from(A,Ain,(LDS_PTR uint32_t *) shp->winA);
  wseq_from(B,Bin,(LDS_PTR uint32_t *) shp->winB);
  GLB_PTR v4i *pool = (GLB_PTR v4i *) uni64((int64_t) G.pool);
  // the level the next cells go to: logical [.., cur_hi) at `cells`; every call starts over in level 0
  GLB_PTR v4i *cells = pool + uni64((int64_t) shp->lvl_off[0]);
  int cur_hi = ARENA_END(0);
  const int maxd = UNI(maxd_in), mida = UNI(mida_in), minp = UNI(minp_in), maxp = UNI(maxp_in), aoff = UNI(aoff_in);
  unsigned long long nwaves = 0, nspill = 0, ncells = 0;
  unsigned int lsum = 0;                   // this lane's snake bases (wave-reduced once, at the end of the call)
  const int ts = TS, path_ave = UNI(G.path_ave), mscore = UNI(G.mscore);
  // `tot` more cells at [avail, avail+tot): when they do not fit the current level, a batch of at most 64 moves to the
  // start of the next level (the skipped tail is never referenced), a larger one may span levels and is stored through
  // ARENA_CELL; either way the levels are taken from the pool first.  Out of line: the common step only compares.
#define ARENA_GROW(tot)                                                                       \
  { int lv_ = ARENA_LEVEL(avail);                                                             \
    if ((tot) <= 64)                                                                          \
      { if (avail + (tot) > ARENA_END(lv_)) { lv_ += 1; avail = ARENA_START(lv_); }           \
        if (!arena_ensure(G,lv_)) BAIL(1)                                                     \
      }                                                                                       \
    else if ((unsigned) avail + (unsigned) (tot) >= (1u << 30) || !arena_ensure(G,ARENA_LEVEL(avail + (tot) - 1))) \
      BAIL(1)                                                                                 \
    cells = pool + uni64((int64_t) shp->lvl_off[lv_]);                                        \
    cur_hi = ARENA_END(lv_);                                                                  \
  }
  const bool force_lds = UNI(G.force_lds) != 0;
  const int VNEW = (S > 0) ? -1 : BIGI;
  int low = UNI(mind), hgh = maxd, dif = 0, cur = 0;
  int more = 1, avail = 0;
  int aclip = (S > 0) ? BIGI : -BIGI;
  int bclip = (S > 0) ? -BIGI : BIGI;
  int besta, bestx, trima, trimx, trimd, trimha, lasta;

  besta = trima = lasta = mida;
  bestx = trimx = (mida+hgh)>>1;
  trimd = 0;
  trimha = 0;

  if (hgh-low+8 >= RC)
    BAIL(2)

  bool regmode = !force_lds && (hgh-low+1 <= REG_MAXW);
  int  kref = 0;
  int      V = VNEW, HA = -1, HM = 0, NA = 0;      // register-mode state of this lane's diagonal
  uint64_t T = PATH_INT;
  if (regmode)
    { const int l0 = (64 - (hgh-low+1)) >> 1;
      kref = (S > 0) ? hgh + l0 : low - l0;
    }

  if (UNI(Ain.p0 < 0)) wseq_load(A,bestx,(S > 0) ? 512 : WINB-512); else wseq_track<S>(A,bestx);
  if (UNI(Bin.p0 < 0)) wseq_load(B,mida-bestx,(S > 0) ? 512 : WINB-512); else wseq_track<S>(B,mida-bestx);
  WAVE_SYNC();

  // ---- wave 0 (align.c:425-512 / 949-1035) ---------------------------------------------------------
  { const int span = hgh-low+1;
    ncells += (unsigned long long) span;
    for (int j0 = 0; j0 < span; j0 += 64)
      { int k; bool act;
        if (regmode) { k = KOF(lane); act = k >= low && k <= hgh; }
        else         { const int j = j0 + lane; act = j < span; k = (S > 0) ? hgh-j : low+j; }
        int x = 0, c = 0, cnt = 0, na = 0, mark0 = 0, hitA = 0, hitB = 0;
        if (act)
          { x = (mida+k)>>1;
            if (S > 0)
              { na = ((x+(ts-aoff))/ts-1)*ts+aoff;
                mark0 = na;
                na += ts;
              }
            else
              { na = ((x+(ts-aoff)-1)/ts-1)*ts+aoff;
                mark0 = x;
              }
            int y = x-k, L;
            if (S > 0)
              { int ra = A.len-x, rb = B.len-y;
                int lim = ra < rb ? ra : rb;
                L = snake<+1>(A,B,x,y,lim);
                if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                x += L;
              }
            else
              { int ra = x, rb = y;
                int lim = ra < rb ? ra : rb;
                L = snake<-1>(A,B,x,y,lim);
                if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                x -= L;
              }
            if (EXT_ACCOUNT) lsum += (unsigned int) L;
            c = (x << 1) - k;
            if (S > 0) cnt = (x >= na) ? (x-na)/ts+1 : 0;
            else       cnt = (x <= na) ? (na-x)/ts+1 : 0;
          }
        int tot, off = wscan_add_excl(act ? 1+cnt : 0,tot);
        if (UNLIKELY(avail + tot > cur_hi))
          ARENA_GROW(tot)
        int ha = -1, hm = 0;
        if (act)
          { int idx = avail + off;
            *ARENA_CELL(pool,idx) = (v4i) { -1,k,0,mark0 };
            ha = idx; hm = mark0;
            for (int q = 0; q < cnt; q++)
              { idx += 1;
                *ARENA_CELL(pool,idx) = (v4i) { ha,k,0,na };
                ha = idx; hm = na;
                na += S*ts;
              }
            if (regmode)
              { V = c; T = PATH_INT; HA = ha; HM = hm; NA = na; }
            else
              { shp->V[0][k & RMASK] = c;
                shp->T[0][k & RMASK] = PATH_INT;
                shp->HA[0][k & RMASK] = ha;
                shp->HM[0][k & RMASK] = hm;
                shp->NA[k & RMASK] = na;
              }
          }
        avail += tot;
        const int cn = act ? ((S > 0) ? c : BIGI - c) : 0;          // non-negative, larger = better
          const int pm = wscan_max_excl_nn(cn);
        bool rec = act && ((S > 0) ? c > besta : c < besta) && cn > pm;
        uint64_t rm = BALLOT(rec);
        if (rm)
          { int l = last_lane(rm);
            besta = trima = lasta = rdlane(c,l);
            bestx = trimx = rdlane(x,l);
            trimha = rdlane(ha,l);
          }
        uint64_t am = BALLOT(hitA), bm = BALLOT(hitB);
        if (am | bm) more = 0;
        if (am) aclip = rdlane(k,last_lane(am));
        if (bm && ((S > 0) ? bclip == -BIGI : bclip == BIGI)) bclip = rdlane(k,first_lane(bm));
      }
  }
  WAVE_SYNC();

  // sequence ends reached: drop the clipped diagonals, go on unless the best point itself sits on an end
  // (align.c:755-780; the "more" tip is only used with reach = 1, which FastGA never sets, FastGA.c:3757)
#define CLIP_UPDATE()                                                                 \
  if (UNLIKELY(more == 0))                                                            \
    { int cb = (S > 0) ? base_at(B,besta-bestx) : base_at(B,besta-bestx-1);           \
      int ca = (S > 0) ? base_at(A,bestx) : base_at(A,bestx-1);                       \
      if (cb != 4 && ca != 4)                                                         \
        more = 1;                                                                     \
      if (S > 0)                                                                      \
        { if (hgh >= aclip) hgh = aclip-1;                                            \
          if (low <= bclip) low = bclip+1;                                            \
          aclip = BIGI; bclip = -BIGI;                                                \
        }                                                                             \
      else                                                                            \
        { if (low <= aclip) low = aclip+1;                                            \
          if (hgh >= bclip) hgh = bclip-1;                                            \
          aclip = -BIGI; bclip = BIGI;                                                \
        }                                                                             \
    }

  CLIP_UPDATE()

  // ---- successive waves (align.c:546-803 / 1067-1323) ----------------------------------------------
  while (more && ((S > 0) ? lasta >= besta - TRIM_MLAG : lasta <= besta + TRIM_MLAG))
    { nwaves += 1;
      const int olow = low, ohgh = hgh;            // the diagonals that hold valid state
      low -= 1;
      hgh += 1;
      const bool newlow = low >= minp, newhgh = hgh <= maxp;
      if (!newlow) low += 1;
      if (!newhgh) hgh -= 1;
      if (UNLIKELY(hgh-low+8 >= RC))
        BAIL(2)
      dif += 1;
      const int width = hgh-low+1;
      if (EXT_ACCOUNT) ncells += (unsigned long long) width;

      // ---- representation switch ----
      if (UNLIKELY(regmode && width > REG_MAXW))
        { const int k = KOF(lane);
          if (k >= olow && k <= ohgh)
            { shp->V[cur][k & RMASK] = V;
              shp->T[cur][k & RMASK] = T;
              shp->HA[cur][k & RMASK] = HA;
              shp->HM[cur][k & RMASK] = HM;
              shp->NA[k & RMASK] = NA;
            }
          regmode = false;
          nspill += 1;
          WAVE_SYNC();
        }
      else if (UNLIKELY(!regmode && width <= REG_BACK && !force_lds))
        { const int l0 = (64 - width) >> 1;
          kref = (S > 0) ? hgh + l0 : low - l0;
          const int k = KOF(lane);
          V = VNEW; HA = -1; HM = 0; NA = 0; T = PATH_INT;
          if (k >= olow && k <= ohgh)
            { V  = shp->V[cur][k & RMASK];
              T  = shp->T[cur][k & RMASK];
              HA = shp->HA[cur][k & RMASK];
              HM = shp->HM[cur][k & RMASK];
              NA = shp->NA[k & RMASK];
            }
          regmode = true;
        }

      if (UNLIKELY((dif & 3) == 0))       // fetches outside the window fall back to HBM, so tracking may lag a few steps
        { wseq_track<S>(A,bestx);
          wseq_track<S>(B,besta-bestx);
        }

      uint64_t anyA = 0, anyB = 0;

      if (LIKELY(regmode))
        { // keep the active lanes inside [1,62]: rotate every register when the wave has drifted
          { const int la = LOF((S > 0) ? hgh : low), lb = LOF((S > 0) ? low : hgh);     // first / last active lane
            if (UNLIKELY(la < 1 || lb > 62))
              { const int want = (64 - (lb-la+1)) >> 1;
                const int delta = want - la;                    // new lane = old lane + delta
                const int srcl = lane - delta;
                V  = __shfl(V,srcl,64);
                HA = __shfl(HA,srcl,64);
                HM = __shfl(HM,srcl,64);
                NA = __shfl(NA,srcl,64);
                uint32_t tlo = (uint32_t) __shfl((int) (uint32_t) T,srcl,64);
                uint32_t thi = (uint32_t) __shfl((int) (uint32_t) (T >> 32),srcl,64);
                T = ((uint64_t) thi << 32) | tlo;
                kref = (S > 0) ? kref + delta : kref - delta;
              }
          }
          WAVE_SYNC();
          const int k = KOF(lane);
          const bool act = k >= low && k <= hgh;
          // new diagonals take the trace-point schedule of their inner neighbour; everything outside is VNEW
          { int na_next = FROM_NEXT(NA,0), na_prev = FROM_PREV(NA,0);
            if (newlow && k == low) NA = (S > 0) ? na_prev : na_next;      // NA[low] = NA[low+1]
            if (newhgh && k == hgh) NA = (S > 0) ? na_next : na_prev;      // NA[hgh] = NA[hgh-1]
            if (!act || (newlow && k == low) || (newhgh && k == hgh))
              V = VNEW;
          }
          int x = 0, c = 0, ha = -1, hm = 0, hitA = 0, hitB = 0, ncreate = 0, na = NA, cross = 0;
          uint64_t b = 0;
          { const int ac = V;
            const int a1 = FROM_NEXT(V,VNEW);              // V[k-S]
            const int a2 = FROM_PREV(V,VNEW);              // V[k+S]
            // furthest of self / next lane (k-S) / previous lane (k+S), ties to self, then k-S (align.c:583-608):
            // selects only, no divergent branches
            const bool p1 = (S > 0) ? ac < a1 : ac > a1;
            const int  m1 = p1 ? a1 : ac;
            const bool p2 = (S > 0) ? m1 < a2 : m1 > a2;
            const uint32_t tlo = (uint32_t) T, thi = (uint32_t) (T >> 32);
            const uint32_t nlo = (uint32_t) FROM_NEXT((int) tlo,0), nhi = (uint32_t) FROM_NEXT((int) thi,0);
            const uint32_t plo = (uint32_t) FROM_PREV((int) tlo,0), phi = (uint32_t) FROM_PREV((int) thi,0);
            const int nha = FROM_NEXT(HA,-1), pha = FROM_PREV(HA,-1);
            const int nhm = FROM_NEXT(HM,0),  phm = FROM_PREV(HM,0);
            if (act)
              { c  = p2 ? a2 + S : (p1 ? a1 + S : ac + 2*S);
                ha = p2 ? pha : (p1 ? nha : HA);
                hm = p2 ? phm : (p1 ? nhm : HM);
                const uint32_t blo = p2 ? plo : (p1 ? nlo : tlo), bhi = p2 ? phi : (p1 ? nhi : thi);
                b = (((uint64_t) bhi << 32) | blo) << 1;
                x = (c+k)>>1;
                int y = x-k, L;
                if (S > 0)
                  { int ra = A.len-x, rb = B.len-y;
                    int lim = ra < rb ? ra : rb;
                    L = snake<+1>(A,B,x,y,lim);
                    if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                    x += L;
                  }
                else
                  { int ra = x, rb = y;
                    int lim = ra < rb ? ra : rb;
                    L = snake<-1>(A,B,x,y,lim);
                    if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                    x -= L;
                  }
                b = (L >= 61) ? ~0ull : ((b << L) | ((1ull << L) - 1));
                if (EXT_ACCOUNT) lsum += (unsigned int) L;
                c = (x << 1) - k;
                { const int dx = (S > 0) ? x-na : na-x, dh = (S > 0) ? hm-na : na-hm;
                  cross = (dx >= 0) ? dx/ts+1 : 0;
                  int skip = (dh >= 0) ? dh/ts+1 : 0;
                  if (skip > cross) skip = cross;
                  ncreate = cross - skip;
                }
              }
          }
          // the trim-table lookups of this lane's history (needed below only when the lane sets a new good best
          // point) are issued now, so that their LDS round trip overlaps the pebble stores and the record scan
          const bool tok = act ? trim_ok(shp,b,mscore) : false;
          int tot = 0, off = 0;
          uint64_t cm = BALLOT(ncreate > 0);
          if (LIKELY(cm != 0))
            { const bool single = (BALLOT(ncreate > 1) == 0);
              if (single)                            // the usual case, one pebble per crossing lane: slots by mbcnt
                { off = (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (cm >> 32),__builtin_amdgcn_mbcnt_lo((uint32_t) cm,0u));
                  tot = __popcll(cm);
                }
              else
                off = wscan_add_excl(ncreate,tot);
              if (UNLIKELY(avail + tot > cur_hi))
                ARENA_GROW(tot)
              if (LIKELY(single))
                { if (ncreate > 0)
                    { const int idx = avail + off;
                      hm = na + S*ts*(cross-1);
                      cells[idx] = (v4i) { ha,k,dif,hm };
                      ha = idx;
                    }
                }
              else if (ncreate > 0)
                { int idx = avail + off;
                  int v = na + S*ts*(cross-ncreate);
                  for (int q = 0; q < ncreate; q++)
                    { *ARENA_CELL(pool,idx) = (v4i) { ha,k,dif,v };
                      ha = idx;
                      hm = v;
                      idx += 1;
                      v += S*ts;
                    }
                }
            }
          if (act)
            { NA = na + S*ts*cross;
              V = c; T = b; HA = ha; HM = hm;
            }
          avail += tot;

          // ordered "new best point" scan (align.c:729-742)
          const int cn = act ? ((S > 0) ? c : BIGI - c) : 0;          // non-negative, larger = better
          const int pm = wscan_max_excl_nn(cn);
          bool rec = act && ((S > 0) ? c > besta : c < besta) && cn > pm;
          uint64_t rm = BALLOT(rec);
          if (LIKELY(rm != 0))
            { int l = last_lane(rm);
              besta = rdlane(c,l);
              bestx = rdlane(x,l);
              int m = __popcll(b & WIN61);
              bool good = rec && m >= path_ave;
              uint64_t gm = BALLOT(good);
              if (LIKELY(gm != 0))
                { lasta = rdlane(c,last_lane(gm));
                  const bool trimok = good && tok;
                  uint64_t tm = BALLOT(trimok);
                  if (LIKELY(tm != 0))
                    { int l2 = last_lane(tm);
                      trima = rdlane(c,l2);
                      trimx = rdlane(x,l2);
                      trimd = dif;
                      trimha = rdlane(ha,l2);
                    }
                }
            }
          uint64_t am = BALLOT(hitA), bm = BALLOT(hitB);
          if (am) aclip = KOF(last_lane(am));
          if (bm) bclip = KOF(first_lane(bm));
          anyA = am; anyB = bm;
          if (anyA | anyB) more = 0;

          CLIP_UPDATE()

          // prune both ends (align.c:782-790)
          { const int n = besta - S*WAVE_LAG;
            const bool inr = k >= low && k <= hgh;
            uint64_t km = BALLOT(inr && ((S > 0) ? (V >= n) : (V <= n)));
            if (km == 0)
              hgh = low-1;
            else
              { const int f = first_lane(km), l = last_lane(km);
                if (S > 0) { hgh = kref - f; low = kref - l; }
                else       { low = kref + f; hgh = kref + l; }
              }
          }
        }
      else
        { // ---------------- RING mode ----------------
          if (lane == 0)
            { if (newlow)
                { shp->NA[low & RMASK] = shp->NA[(low+1) & RMASK]; shp->V[cur][low & RMASK] = VNEW; }
              if (newhgh)
                { shp->NA[hgh & RMASK] = shp->NA[(hgh-1) & RMASK]; shp->V[cur][hgh & RMASK] = VNEW; }
              shp->V[cur][(hgh+1) & RMASK] = shp->V[cur][(low-1) & RMASK] = VNEW;
            }
          WAVE_SYNC();

          const int span = hgh-low+1;
          const int nxt = cur^1;
          for (int j0 = 0; j0 < span; j0 += 64)
            { const int j = j0 + lane;
              const bool act = j < span;
              const int k = (S > 0) ? hgh-j : low+j;
              int x = 0, c = 0, ha = -1, hm = 0, hitA = 0, hitB = 0, ncreate = 0, na = 0, cross = 0;
              uint64_t b = 0;
              if (act)
                { int ac = shp->V[cur][k & RMASK];
                  int a1 = shp->V[cur][(k-S) & RMASK];
                  int a2 = shp->V[cur][(k+S) & RMASK];
                  int src;
                  if (S > 0)
                    { if (ac < a1) src = (a1 < a2) ? k+S : k-S;
                      else         src = (ac < a2) ? k+S : k;
                    }
                  else
                    { if (ac > a1) src = (a1 > a2) ? k+S : k-S;
                      else         src = (ac > a2) ? k+S : k;
                    }
                  c  = (src == k) ? ac + 2*S : ((src == k-S) ? a1 : a2) + S;
                  b  = shp->T[cur][src & RMASK];
                  ha = shp->HA[cur][src & RMASK];
                  hm = shp->HM[cur][src & RMASK];
                  b <<= 1;
                  x = (c+k)>>1;
                  int y = x-k, L;
                  if (S > 0)
                    { int ra = A.len-x, rb = B.len-y;
                      int lim = ra < rb ? ra : rb;
                      L = snake<+1>(A,B,x,y,lim);
                      if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                      x += L;
                    }
                  else
                    { int ra = x, rb = y;
                      int lim = ra < rb ? ra : rb;
                      L = snake<-1>(A,B,x,y,lim);
                      if (L == rb) hitB = 1; else if (L == ra) hitA = 1;
                      x -= L;
                    }
                  if (L > 0)
                    b = (L >= 61) ? ~0ull : ((b << L) | ((1ull << L) - 1));
                  if (EXT_ACCOUNT) lsum += (unsigned int) L;
                  c = (x << 1) - k;
                  na = shp->NA[k & RMASK];
                  { const int dx = (S > 0) ? x-na : na-x, dh = (S > 0) ? hm-na : na-hm;
                    cross = (dx >= 0) ? dx/ts+1 : 0;
                    int skip = (dh >= 0) ? dh/ts+1 : 0;
                    if (skip > cross) skip = cross;
                    ncreate = cross - skip;
                  }
                }
              int tot = 0, off = 0;
              uint64_t cm = BALLOT(ncreate > 0);
              if (cm)
                { off = wscan_add_excl(ncreate,tot);
                  if (UNLIKELY(avail + tot > cur_hi))
                    ARENA_GROW(tot)
                }
              if (act)
                { if (ncreate > 0)
                    { int idx = avail + off;
                      int v = na + S*ts*(cross-ncreate);
                      for (int q = 0; q < ncreate; q++)
                        { *ARENA_CELL(pool,idx) = (v4i) { ha,k,dif,v };
                          ha = idx;
                          hm = v;
                          idx += 1;
                          v += S*ts;
                        }
                    }
                  if (cross > 0)
                    shp->NA[k & RMASK] = na + S*ts*cross;
                  shp->V[nxt][k & RMASK] = c;
                  shp->T[nxt][k & RMASK] = b;
                  shp->HA[nxt][k & RMASK] = ha;
                  shp->HM[nxt][k & RMASK] = hm;
                }
              avail += tot;

              const int cn = act ? ((S > 0) ? c : BIGI - c) : 0;          // non-negative, larger = better
          const int pm = wscan_max_excl_nn(cn);
              bool rec = act && ((S > 0) ? c > besta : c < besta) && cn > pm;
              uint64_t rm = BALLOT(rec);
              if (rm)
                { int l = last_lane(rm);
                  besta = rdlane(c,l);
                  bestx = rdlane(x,l);
                  int m = __popcll(b & WIN61);
                  bool good = rec && m >= path_ave;
                  uint64_t gm = BALLOT(good);
                  if (gm)
                    { lasta = rdlane(c,last_lane(gm));
                      const bool trimok = good && trim_ok(shp,b,mscore);
                      uint64_t tm = BALLOT(trimok);
                      if (tm)
                        { int l2 = last_lane(tm);
                          trima = rdlane(c,l2);
                          trimx = rdlane(x,l2);
                          trimd = dif;
                          trimha = rdlane(ha,l2);
                        }
                    }
                }
              uint64_t am = BALLOT(hitA), bm = BALLOT(hitB);
              if (am) aclip = rdlane(k,last_lane(am));
              if (bm && !anyB) bclip = rdlane(k,first_lane(bm));
              anyA |= am; anyB |= bm;
            }
          if (anyA | anyB) more = 0;
          cur = nxt;
          WAVE_SYNC();

          CLIP_UPDATE()

          // prune both ends (align.c:782-790)
          { const int n = besta - S*WAVE_LAG;
            int nh = low-1, nl = hgh+1;
            const int sp = hgh-low+1;
            for (int j0 = 0; j0 < sp; j0 += 64)
              { int k = low + j0 + lane;
                bool keep = false;
                if (k <= hgh)
                  { int v = shp->V[cur][k & RMASK];
                    keep = (S > 0) ? (v >= n) : (v <= n);
                  }
                uint64_t km = BALLOT(keep);
                if (km)
                  { int f = low + j0 + first_lane(km), l = low + j0 + last_lane(km);
                    if (f < nl) nl = f;
                    if (l > nh) nh = l;
                  }
              }
            if (nh < nl)          // nothing survives: the reference leaves hgh < low
              hgh = low-1;
            else
              { hgh = nh; low = nl; }
          }
        }
    }

  PF.t_steps += clock64() - tstart;
  PF.nsteps += nspill;
  PF.ncells += ncells;
  { int tot;                                // one wave reduction per call (lsum < 2^31: a call touches < 2^31 bases per lane)
    wscan_add_excl((int) lsum,tot);
    PF.nbases += (unsigned long long) (unsigned int) tot;
  }
  nwaves_out += nwaves;
  WIN_BACK()
  ext_unwind<S>(G,trace,tcap,P,PF,mida,aoff,trima,trimx,trimd,trimha,mind);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// Local_Alignment (align.c:1423-1576), wave-uniform
// ---------------------------------------------------------------------------------------------------
__device__ int local_alignment(const ext_args &G, LDS_PTR ext_shared *sh, uint16_t *trace, int tcap,
                               ext_seq &A, ext_seq &B, int acomp,
                               int low, int hgh, int anti, int lbord, int hbord,
                               ext_state &P, unsigned long long &nwaves, ext_prof &PF, int selfie = 0)
{ int minp, maxp, aoff, st;
  P.tpos = tcap/2;
  P.tlen = 0;
  while (((anti-hgh)>>1) < 0)
    hgh -= 1;
  // selfie: the reference's `aseq == bseq` rule (align.c:1461-1481) -- never the case in FastGA's own calls, which load
  // the two contigs into separate buffers; the exact-signature shim passes it through
  minp = (lbord < 0) ? ((selfie && low >= 0) ? 1 : -BIGI) : low-lbord;
  maxp = (hbord < 0) ? ((selfie && hgh <= 0) ? -1 : BIGI) : hgh+hbord;
  aoff = acomp ? A.len % TS : 0;

  if ((st = ext_wave<+1>(G,sh,trace,tcap,A,B,P,low,hgh,anti,minp,maxp,aoff,nwaves,PF)) != 0) return st;
  int fshort = ((P.aepos + P.bepos) - anti < DUB_TRIM);
  { int l2 = low;
    if ((st = ext_wave<-1>(G,sh,trace,tcap,A,B,P,l2,low,anti,minp,maxp,aoff,nwaves,PF)) != 0) return st;
  }
  int rshort = (anti - (P.abpos + P.bbpos) < DUB_TRIM);
  if (fshort)
    { if (rshort)
        { P.aepos = P.abpos = (P.abpos+P.aepos)>>1;
          P.bepos = P.bbpos = (P.bbpos+P.bepos)>>1;
          P.tlen = 0;
        }
      else
        { low  = P.abpos - P.bbpos;
          anti = P.abpos + P.bbpos;
          P.tlen = 0;
          if ((st = ext_wave<+1>(G,sh,trace,tcap,A,B,P,low,low,anti,minp,maxp,aoff,nwaves,PF)) != 0) return st;
        }
    }
  else if (rshort)
    { low  = P.aepos - P.bepos;
      anti = P.aepos + P.bepos;
      P.tlen = 0;
      P.diffs = 0;
      if ((st = ext_wave<-1>(G,sh,trace,tcap,A,B,P,low,low,anti,minp,maxp,aoff,nwaves,PF)) != 0) return st;
    }
  if (acomp)
    { int i = P.abpos; P.abpos = A.len - P.aepos; P.aepos = A.len - i;
      i = P.bbpos;     P.bbpos = B.len - P.bepos; P.bepos = B.len - i;
      // the trace pairs are reversed when they are copied out (see emit)
    }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// kernel: persistent wavefronts pull units from a queue (longest estimated first)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64)
void extend_kernel(ext_args G)
{ LDS_PTR ext_shared *sh = (LDS_PTR ext_shared *) &ext_lds;
  trim_fill(sh,G.mscore);
  __syncthreads();
  const int lane = threadIdx.x;
  // arena level 0 and -- per unit, sized by its A contig -- the trace scratch come from the launch's pool
  if (lane == 0) sh->nlev = 0;
  __syncthreads();
  bool pool_ok = arena_ensure(G,0);
  uint16_t *trace = NULL;
  int tcap = 0;                             // uint16 elements of the scratch; trace[0] of a call sits in the middle
  unsigned long long ncalls = 0, nwaves = 0;
  ext_prof PF; PF.t_steps = PF.t_unwind = PF.t_total = PF.nsteps = 0; PF.ncells = PF.nbases = 0;
  const unsigned long long tk0 = clock64();

  while (1)
    { int ui = 0;
      if (lane == 0)
        ui = atomicAdd(G.next,1);
      ui = __shfl(ui,0,64);
      if (ui >= G.nunits)
        break;
      const int u = G.order[ui];
      const fga_unit U = G.units[u];
      const int comp = U.comp;
      const int ctg1 = G.permA[U.actg], ctg2 = G.permB[U.bctg];
      ext_seq A, B;
      A.len = (int) G.clenA[ctg1];
      B.len = (int) G.clenB[ctg2];
      A.win = (LDS_PTR uint32_t *) sh->winA; A.p0 = -1; A.w0 = 0;
      B.win = (LDS_PTR uint32_t *) sh->winB; B.p0 = -1; B.w0 = 0;
      B.img = G.imgB; B.base = (G.padB + G.boffB[ctg2]) * 4; B.bsh = (int) (B.base & 15);
      if (comp)
        { A.img = G.imgAr; A.base = (G.padA + G.boffA[ctg1]) * 4; }
      else
        { A.img = G.imgA;  A.base = (G.padA + G.boffA[ctg1]) * 4; }
      A.bsh = (int) (A.base & 15);
      const int64_t mlen = (int64_t) A.len + B.len;
      const int self = G.self && ctg1 == ctg2 && !comp;

      int64_t alast = -1;
      int seq = 0;
      ext_state P;
      P.abpos = P.bbpos = P.aepos = P.bepos = P.diffs = P.tlen = 0; P.tpos = 0;
      { // forward pairs are stored down from the top, reverse pairs below the middle: 4 (len/100 + 2) elements each
        const int need = 8*(A.len/TS + 8) + 64;
        if (pool_ok && need > tcap)
          { const int ncap = need > 2*tcap ? need : 2*tcap;
            const long long at = pool_take(G,((long long) ncap*2 + 15) / 16 + 1);
            if (at < 0) pool_ok = false;
            else { trace = (uint16_t *) (G.pool + at); tcap = ncap; }
          }
        if (!pool_ok)
          { if (lane == 0)
              atomicMax(G.counters+4,1ull);
            break;
          }
      }
      for (int hi = 0; hi < U.nhits; hi++)
        { const fga_hit H = G.hits[U.first_hit + hi];
          int dgmin = H.dgmin, dgmax = H.dgmax;
          int64_t alow = H.alow, ahgh = H.ahgh, amid, eant;
          if (ahgh <= alast)
            continue;
          if (alow < alast)
            alow = alast;
          ahgh -= BUCK_ANTI;
          do
            { amid = alow + BUCK_ANTI;
              if (amid > ahgh)
                { amid = ahgh;
                  if (amid + dgmin < 0)
                    { dgmin = (int) -amid;
                      if (dgmin > dgmax)
                        break;
                    }
                }
              int st = 0, called = 1;
              if (self)
                { if (dgmin > 0)
                    st = local_alignment(G,sh,trace,tcap,A,B,comp,dgmin,dgmax,(int) amid,dgmin-1,-1,P,nwaves,PF);
                  else if (dgmax < 0)
                    st = local_alignment(G,sh,trace,tcap,A,B,comp,dgmin,dgmax,(int) amid,-1,-(dgmax+1),P,nwaves,PF);
                  else
                    { P.abpos = P.aepos = 0; called = 0; }
                }
              else
                st = local_alignment(G,sh,trace,tcap,A,B,comp,dgmin,dgmax,(int) amid,-1,-1,P,nwaves,PF);
              ncalls += called;
              if (st != 0)
                { if (lane == 0)
                    atomicMax(G.counters+4,(unsigned long long) st);
                  goto unit_done;
                }
              const int rlen = P.aepos - P.abpos;
              if (rlen >= G.aln_min && G.aln_rate*rlen >= (double) P.diffs)
                { // emit: Overlap record + trace bytes (reversed pair order for the complement pass)
                  unsigned long long ai = 0, to = 0;
                  if (lane == 0)
                    { ai = atomicAdd(G.counters+0,1ull);
                      to = atomicAdd(G.counters+1,(unsigned long long) P.tlen);
                    }
                  ai = __shfl(ai,0,64);
                  to = __shfl(to,0,64);
                  if ((int64_t) ai < G.aln_cap && (int64_t) (to + P.tlen) <= G.tbytes_cap)
                    { if (lane == 0)
                        { fga_aln R;
                          R.tlen = P.tlen; R.diffs = P.diffs;
                          R.abpos = P.abpos; R.bbpos = P.bbpos; R.aepos = P.aepos; R.bepos = P.bepos;
                          R.flags = comp ? 1u : 0u;
                          R.aread = ctg1; R.bread = ctg2;
                          R.unit = u; R.seq = seq;
                          R.toff = (int64_t) to;
                          G.alns[ai] = R;
                        }
                      const uint16_t *src = trace + P.tpos;
                      const int np = P.tlen >> 1;
                      for (int q = lane; q < np; q += 64)
                        { int sp = comp ? (np-1-q) : q;
                          G.tbytes[to + 2*q]   = (uint8_t) src[2*sp];
                          G.tbytes[to + 2*q+1] = (uint8_t) src[2*sp+1];
                        }
                    }
                  seq += 1;
                }
              if (comp)
                eant = mlen - ((int64_t) P.abpos + P.bbpos);
              else
                eant = (int64_t) P.aepos + P.bepos;
              if (eant <= alow)
                alow = amid;
              else
                alow = eant;
            }
          while (alow < ahgh);
          alast = alow;
        }
    unit_done: ;
    }
  if (lane == 0)
    { atomicAdd(G.counters+2,ncalls);
      atomicAdd(G.counters+3,nwaves);
      atomicMax(G.counters+5,PF.t_steps);
      atomicMax(G.counters+6,PF.t_unwind);
      atomicMax(G.counters+7,clock64()-tk0);
      atomicAdd(G.counters+8,PF.t_steps);
      atomicAdd(G.counters+9,PF.t_unwind);
      atomicMax(G.counters+10,nwaves);
      atomicAdd(G.counters+11,PF.nsteps);
      atomicAdd(G.counters+12,PF.ncells);
      atomicAdd(G.counters+13,PF.nbases);
    }
}

// ---------------------------------------------------------------------------------------------------
// one Local_Alignment call on two packed sequences: the device side of the exact-signature shim (fga_shim_Local_Alignment)
// ---------------------------------------------------------------------------------------------------
struct la_one_args
  { ext_args G;
    int alen, blen, acomp, selfie, low, hgh, anti, lbord, hbord;
    int *out;                  // status, abpos, bbpos, aepos, bepos, diffs, tlen
    uint16_t *trace_out; int trace_cap;
  };

__global__ __launch_bounds__(64)
void local_alignment_one_kernel(la_one_args L)
{ LDS_PTR ext_shared *sh = (LDS_PTR ext_shared *) &ext_lds;
  const ext_args &G = L.G;
  trim_fill(sh,G.mscore);
  const int lane = threadIdx.x;
  if (lane == 0) sh->nlev = 0;
  __syncthreads();
  int st = arena_ensure(G,0) ? 0 : 1;
  ext_seq A, B;
  A.len = L.alen; B.len = L.blen;
  A.win = (LDS_PTR uint32_t *) sh->winA; A.p0 = -1; A.w0 = 0;
  B.win = (LDS_PTR uint32_t *) sh->winB; B.p0 = -1; B.w0 = 0;
  A.img = G.imgA; A.base = G.padA * 4; A.bsh = (int) (A.base & 15);
  B.img = G.imgB; B.base = G.padB * 4; B.bsh = (int) (B.base & 15);
  const int maxlen = L.alen > L.blen ? L.alen : L.blen;
  const int tcap = 8*(maxlen/TS + 8) + 64;
  uint16_t *trace = NULL;
  if (st == 0)
    { const long long at = pool_take(G,((long long) tcap*2 + 15) / 16 + 1);
      if (at < 0) st = 1; else trace = (uint16_t *) (G.pool + at);
    }
  ext_state P;
  P.abpos = P.bbpos = P.aepos = P.bepos = P.diffs = P.tlen = 0; P.tpos = 0;
  unsigned long long nwaves = 0;
  ext_prof PF; PF.t_steps = PF.t_unwind = PF.t_total = PF.nsteps = 0; PF.ncells = PF.nbases = 0;
  if (st == 0)
    st = local_alignment(G,sh,trace,tcap,A,B,L.acomp,L.low,L.hgh,L.anti,L.lbord,L.hbord,P,nwaves,PF,L.selfie);
  if (st == 0 && P.tlen > L.trace_cap)
    st = 3;
  if (st == 0)
    { const uint16_t *src = trace + P.tpos;
      const int np = P.tlen >> 1;
      for (int q = lane; q < np; q += 64)             // pairs come out reversed for a complemented A (align.c:1534-1555)
        { const int sp = L.acomp ? (np-1-q) : q;
          L.trace_out[2*q]   = src[2*sp];
          L.trace_out[2*q+1] = src[2*sp+1];
        }
    }
  if (lane == 0)
    { L.out[0] = st;
      L.out[1] = P.abpos; L.out[2] = P.bbpos; L.out[3] = P.aepos; L.out[4] = P.bepos; L.out[5] = P.diffs; L.out[6] = P.tlen;
    }
}

// ---------------------------------------------------------------------------------------------------
// reverse-complement image of a packed genome (Complement_Seq of every contig, align.c:4082-4097)
// ---------------------------------------------------------------------------------------------------
__global__ void revcomp_kernel(const uint8_t *fwd, uint8_t *rc, const int64_t *boff, const int64_t *clen,
                               int nctg, int64_t pad)
{ const int c = blockIdx.y;
  if (c >= nctg) return;
  const int64_t len = clen[c], nb = (len+3) >> 2;
  const uint8_t *src = fwd + pad + boff[c];
  uint8_t *dst = rc + pad + boff[c];
  for (int64_t ob = (int64_t) blockIdx.x*blockDim.x + threadIdx.x; ob < nb; ob += (int64_t) gridDim.x*blockDim.x)
    { uint8_t v = 0;
      for (int q = 0; q < 4; q++)
        { int64_t x = ob*4 + q;             // position in the complemented contig
          if (x < len)
            { int64_t y = len-1-x;
              int bse = (src[y >> 2] >> (2*(y & 3))) & 3;
              v |= (uint8_t) ((3-bse) << (2*q));
            }
        }
      dst[ob] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#define IMG_PAD 4096    // bytes of zero padding before and after a genome image (>= one LDS window)

extern "C" int fga_dgenome_upload(fga_dev *dev, const fga_gdb *G, const int *perm, int nperm, int want_revcomp,
                                  fga_dgenome **out)
{ *out = NULL;
  FGA_HIP(hipSetDevice(dev->device));
  fga_dgenome *D = (fga_dgenome *) calloc(1,sizeof(fga_dgenome));
  if (D == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  D->dev = dev;
  D->nctg = G->ncontig;
  D->nperm = nperm;
  D->pad = IMG_PAD;
  D->maxctg = G->maxctg;
  const size_t bytes = (size_t) G->bpslen + 2*IMG_PAD + 16;
  std::vector<int64_t> boff(G->ncontig), clen(G->ncontig);
  for (int c = 0; c < G->ncontig; c++)
    { boff[c] = G->contigs[c].boff; clen[c] = G->contigs[c].clen; }
  hipError_t e;
  if ((e = hipMalloc(&D->img,bytes)) != hipSuccess ||
      (e = hipMalloc(&D->boff,sizeof(int64_t)*G->ncontig)) != hipSuccess ||
      (e = hipMalloc(&D->clen,sizeof(int64_t)*G->ncontig)) != hipSuccess ||
      (e = hipMalloc(&D->perm,sizeof(int)*(nperm > 0 ? nperm : 1))) != hipSuccess ||
      (want_revcomp && (e = hipMalloc(&D->img_rc,bytes)) != hipSuccess))
    { fga_set_error("fga_dgenome_upload: device allocation failed: %s",hipGetErrorString(e));
      hipFree(D->img); hipFree(D->boff); hipFree(D->clen); hipFree(D->perm); hipFree(D->img_rc); free(D);
      return 1;
    }
  hipMemset(D->img,0,bytes);
  hipMemcpy(D->img + IMG_PAD,G->bps,G->bpslen,hipMemcpyHostToDevice);
  hipMemcpy(D->boff,boff.data(),sizeof(int64_t)*G->ncontig,hipMemcpyHostToDevice);
  hipMemcpy(D->clen,clen.data(),sizeof(int64_t)*G->ncontig,hipMemcpyHostToDevice);
  D->hclen = (int64_t *) malloc(sizeof(int64_t)*(G->ncontig > 0 ? G->ncontig : 1));
  if (D->hclen != NULL)
    memcpy(D->hclen,clen.data(),sizeof(int64_t)*G->ncontig);
  if (nperm > 0)
    hipMemcpy(D->perm,perm,sizeof(int)*nperm,hipMemcpyHostToDevice);
  if (want_revcomp)
    { hipMemset(D->img_rc,0,bytes);
      dim3 grid(256,G->ncontig);
      hipLaunchKernelGGL(revcomp_kernel,grid,dim3(256),0,dev->stream,D->img,D->img_rc,D->boff,D->clen,
                         G->ncontig,(int64_t) IMG_PAD);
    }
  e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("fga_dgenome_upload: %s",hipGetErrorString(e));
      hipFree(D->img); hipFree(D->boff); hipFree(D->clen); hipFree(D->perm); hipFree(D->img_rc); free(D->hclen); free(D);
      return 1;
    }
  *out = D;
  return 0;
}

extern "C" void fga_dgenome_free(fga_dgenome *D)
{ if (D == NULL) return;
  hipSetDevice(D->dev->device);
  hipFree(D->img); hipFree(D->img_rc); hipFree(D->boff); hipFree(D->clen); hipFree(D->perm);
  free(D->hclen);
  free(D);
}

extern "C" int fga_extend(fga_dev *dev, const fga_dgenome *GA, const fga_dgenome *GB, const fga_hits *H,
                          const fga_extend_params *prm, fga_alns **out)
{ *out = NULL;
  FGA_HIP(hipSetDevice(dev->device));
  fga_alns *R = (fga_alns *) calloc(1,sizeof(fga_alns));
  if (R == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  dev->last_ms[FGA_STAGE_EXTEND] = 0.f;
  if (H->nunits == 0)
    { *out = R;
      return 0;
    }
  if (GA->img_rc == NULL)
    { fga_set_error("fga_extend: genome 1 was uploaded without its reverse-complement image");
      free(R);
      return 1;
    }

  // units by decreasing estimated work (sum of hit box lengths): longest first
  std::vector<int> order(H->nunits);
  { std::vector<uint64_t> w((size_t) H->nunits);
    int64_t smax = 0;
    for (int64_t u = 0; u < H->nunits; u++)
      { int64_t s = 0;
        for (int q = 0; q < H->units[u].nhits; q++)
          { const fga_hit &h = H->hits[H->units[u].first_hit + q];
            s += (h.ahgh - h.alow) + 1000;
          }
        if (s < 0) s = 0;
        w[(size_t) u] = (uint64_t) s;
        if (s > smax) smax = s;
      }
    for (int64_t u = 0; u < H->nunits; u++)            // decreasing work, ties by unit index (the radix order is stable)
      w[(size_t) u] = (uint64_t) smax - w[(size_t) u];
    std::vector<int64_t> ord;
    fga_radix_order(w.data(),H->nunits,ord);
    for (int64_t u = 0; u < H->nunits; u++) order[u] = (int) ord[(size_t) u];
  }

  // resident single-wavefront workgroups: the LDS block of a wavefront (26.5 KB) allows six per CU.  With few units the
  // run time is the longest unit's serial chain, which is fastest with four (it shares its CU's issue slots, LDS and L1
  // with fewer neighbours: 80.7 vs 83.9 ms on the bench pair); with many units throughput wins (1 M units: 792 -> 647 ms)
  int nwg = dev->ncu * (H->nunits >= 16*(int64_t) dev->ncu*4 ? 6 : 4);
  { const char *ev = getenv("FGA_EXTEND_WGS");
    if (ev != NULL && atoi(ev) > 0) nwg = atoi(ev);
  }
  if (nwg > H->nunits) nwg = (int) H->nunits;
  const int64_t maxa = GA->maxctg > GB->maxctg ? GA->maxctg : GB->maxctg;
  // Scratch and output are sized by what the hits suggest, not by the worst case times the number of wavefronts; the
  // kernel counts what it would have needed, so a launch that runs out is repeated once with exactly that
  // (prm->cell_cap: pool cells; prm->aln_cap / prm->trace_cap: output records / trace bytes).
  int64_t span = 0;
  for (int64_t h = 0; h < H->nhits; h++)
    span += (H->hits[h].ahgh - H->hits[h].alow) / 2 + 200;
  int64_t pool_cells = prm->cell_cap > 0 ? prm->cell_cap
                                         : (int64_t) nwg*((1 << ARENA_L0) + 8*(maxa/prm->tspace + 72)/8 + 64) + 2*24*(span/prm->tspace)
                                           + ((int64_t) 64 << 20);
  int64_t aln_cap   = prm->aln_cap   > 0 ? prm->aln_cap   : 4*H->nhits + 1024;
  int64_t tb_cap    = prm->trace_cap > 0 ? prm->trace_cap : 4*(span/prm->tspace) + 64*aln_cap + (1 << 20);
  { size_t fr = 0, tot = 0;                  // never ask for more than half of what is free
    if (hipMemGetInfo(&fr,&tot) == hipSuccess && fr > 0)
      { const int64_t lim = (int64_t) (fr/2) / (int64_t) sizeof(int4);
        if (pool_cells > lim) pool_cells = lim;
      }
  }

  ext_args A;
  memset(&A,0,sizeof(A));
  A.imgA = (const uint32_t *) GA->img; A.imgAr = (const uint32_t *) GA->img_rc; A.imgB = (const uint32_t *) GB->img;
  A.boffA = GA->boff; A.boffB = GB->boff; A.clenA = GA->clen; A.clenB = GB->clen;
  A.permA = GA->perm; A.permB = GB->perm; A.padA = GA->pad; A.padB = GB->pad;
  A.nunits = (int) H->nunits;
  A.tspace = prm->tspace; A.path_ave = prm->path_ave; A.self = prm->self;
  A.aln_min = prm->aln_min; A.aln_rate = prm->aln_rate;
  A.mscore = prm->score[0x7fff] / 15;      // SCORE[all matches] = 15 * mscore
  A.force_lds = getenv("FGA_EXTEND_FORCE_LDS") != NULL;

  fga_unit *d_units = NULL; fga_hit *d_hits = NULL; int *d_order = NULL, *d_next = NULL;
  int16_t *d_tab = NULL;
  unsigned long long *d_cnt = NULL;
  hipError_t e;
  if ((e = hipMalloc(&d_units,sizeof(fga_unit)*H->nunits)) != hipSuccess ||
      (e = hipMalloc(&d_hits,sizeof(fga_hit)*(H->nhits+1))) != hipSuccess ||
      (e = hipMalloc(&d_order,sizeof(int)*H->nunits)) != hipSuccess ||
      (e = hipMalloc(&d_next,sizeof(int))) != hipSuccess ||
      (e = hipMalloc(&d_tab,sizeof(int16_t)*2*32768)) != hipSuccess ||
      (e = hipMalloc(&d_cnt,sizeof(unsigned long long)*32)) != hipSuccess)
    { fga_set_error("fga_extend: device allocation failed: %s",hipGetErrorString(e));
      goto fail;
    }
  hipMemcpy(d_units,H->units,sizeof(fga_unit)*H->nunits,hipMemcpyHostToDevice);
  hipMemcpy(d_hits,H->hits,sizeof(fga_hit)*H->nhits,hipMemcpyHostToDevice);
  hipMemcpy(d_order,order.data(),sizeof(int)*H->nunits,hipMemcpyHostToDevice);
  hipMemcpy(d_tab,prm->table,sizeof(int16_t)*32768,hipMemcpyHostToDevice);
  hipMemcpy(d_tab+32768,prm->score,sizeof(int16_t)*32768,hipMemcpyHostToDevice);
  A.units = d_units; A.hits = d_hits; A.order = d_order; A.next = d_next;
  A.table = d_tab; A.score = d_tab+32768; A.counters = d_cnt;

  for (int attempt = 0; ; attempt++)
    { A.pool_cells = pool_cells; A.aln_cap = aln_cap; A.tbytes_cap = tb_cap;
      A.pool   = (int4 *)     fga_dev_acquire(dev,SLOT_CELLS,sizeof(int4)*((size_t) pool_cells + 128));   // + a window of padding
      A.alns   = (fga_aln *)  fga_dev_acquire(dev,SLOT_ALNS,sizeof(fga_aln)*(size_t) aln_cap);
      A.tbytes = (uint8_t *)  fga_dev_acquire(dev,SLOT_TBYTES,(size_t) tb_cap);
      if (A.pool == NULL || A.alns == NULL || A.tbytes == NULL)
        { fga_set_error("fga_extend: device allocation failed (%lld MB trace-point pool, %lld MB trace bytes)",
                        (long long) (sizeof(int4)*(size_t) pool_cells >> 20),(long long) (tb_cap >> 20));
          goto fail;
        }
      A.pool_next = d_cnt + 16;
      hipMemsetAsync(d_next,0,sizeof(int),dev->stream);
      hipMemsetAsync(d_cnt,0,sizeof(unsigned long long)*32,dev->stream);

      hipEventRecord(dev->ev0,dev->stream);
      hipLaunchKernelGGL(extend_kernel,dim3(nwg),dim3(64),0,dev->stream,A);
      hipEventRecord(dev->ev1,dev->stream);
      e = hipStreamSynchronize(dev->stream);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess)
        { fga_set_error("fga_extend: kernel failed: %s",hipGetErrorString(e));
          goto fail;
        }
      hipEventElapsedTime(&dev->last_ms[FGA_STAGE_EXTEND],dev->ev0,dev->ev1);
      unsigned long long hc[32];
      hipMemcpy(hc,d_cnt,sizeof(hc),hipMemcpyDeviceToHost);
      if (getenv("FGA_EXTEND_PROFILE") != NULL)
        fprintf(stderr,"extend profile: max per wavefront: steps %.2f Mcyc, unwind %.2f Mcyc, total %.2f Mcyc, waves %llu; "
                       "sum: steps %.1f Mcyc unwind %.1f Mcyc; kernel %.2f ms, %d workgroups, %lld units, %llu register->ring spills, "
                       "pool %.1f of %.1f MB\n",
                hc[5]*1e-6,hc[6]*1e-6,hc[7]*1e-6,hc[10],hc[8]*1e-6,hc[9]*1e-6,dev->last_ms[FGA_STAGE_EXTEND],nwg,
                (long long) H->nunits,hc[11],hc[16]*16e-6,pool_cells*16e-6);

      R->naln = (int64_t) hc[0]; R->ntrace = (int64_t) hc[1]; R->ncalls = (int64_t) hc[2]; R->nwaves = (int64_t) hc[3];
      R->ncells = (int64_t) hc[12]; R->nbases = (int64_t) hc[13];
      // wavefronts busy on average = wave time spent in steps + unwinds, over the longest wavefront's lifetime
      R->busy_waves = hc[7] > 0 ? (double) (hc[8] + hc[9]) / (double) hc[7] : 0.;
      const bool pool_out = (hc[4] == 1), out_full = (R->naln > aln_cap || R->ntrace > tb_cap);
      if (hc[4] > 1)
        { fga_set_error("fga_extend: wave wider than the LDS ring (512 diagonals)");
          goto fail;
        }
      if (!pool_out && !out_full)
        break;
      fga_dev_release(dev,SLOT_CELLS,A.pool); fga_dev_release(dev,SLOT_ALNS,A.alns); fga_dev_release(dev,SLOT_TBYTES,A.tbytes);
      A.pool = NULL; A.alns = NULL; A.tbytes = NULL;
      if (attempt >= 3 || prm->cell_cap > 0 || prm->aln_cap > 0 || prm->trace_cap > 0)
        { fga_set_error("fga_extend: %s",pool_out ? "trace-point pool exhausted (raise cell_cap)"
                                                  : "output buffers too small (raise aln_cap / trace_cap)");
          goto fail;
        }
      // the counters kept counting: the demand of the part that ran is known.  A pool that ran out stopped wavefronts
      // early, so it is at least doubled
      if (pool_out)  pool_cells = 2*pool_cells > (int64_t) hc[16] ? 2*pool_cells : (int64_t) hc[16] + pool_cells;
      if (R->naln > aln_cap)  aln_cap = R->naln + R->naln/8 + 1024;
      if (R->ntrace > tb_cap) tb_cap = R->ntrace + R->ntrace/8 + (1 << 20);
    }
  R->alns = (fga_aln *) malloc(sizeof(fga_aln)*(R->naln+1));
  R->tbytes = (uint8_t *) malloc(R->ntrace+16);
  if (R->alns == NULL || R->tbytes == NULL)
    { fga_set_error("out of memory");
      goto fail;
    }
  if (R->naln > 0)
    hipMemcpy(R->alns,A.alns,sizeof(fga_aln)*R->naln,hipMemcpyDeviceToHost);
  if (R->ntrace > 0)
    hipMemcpy(R->tbytes,A.tbytes,R->ntrace,hipMemcpyDeviceToHost);
  hipFree(d_units); hipFree(d_hits); hipFree(d_order); hipFree(d_next); hipFree(d_tab); hipFree(d_cnt);
  fga_dev_release(dev,SLOT_CELLS,A.pool);
  fga_dev_release(dev,SLOT_ALNS,A.alns); fga_dev_release(dev,SLOT_TBYTES,A.tbytes);
  *out = R;
  return 0;

fail:
  hipFree(d_units); hipFree(d_hits); hipFree(d_order); hipFree(d_next); hipFree(d_tab); hipFree(d_cnt);
  fga_dev_release(dev,SLOT_CELLS,A.pool);
  fga_dev_release(dev,SLOT_ALNS,A.alns); fga_dev_release(dev,SLOT_TBYTES,A.tbytes);
  free(R->alns); free(R->tbytes); free(R);
  return 1;
}

// ---------------------------------------------------------------------------------------------------
// exact-signature shims of the reference's module seams: Local_Alignment with New/Free_Work_Data and
// New/Free_Align_Spec (align.h:166-168, 196-198, 235-236).  They let a maintainer swap ONE call inside the unmodified
// pipeline (FastGA.c:3247-3260) or A/B it against align.c: same argument meaning, same result fields, the trace lives in
// storage of the Work_Data and is overwritten by the next call, 1 is returned on failure (message: fga_last_error).
// Every call packs the two NUMERIC sequences to 2 bits, uploads them and runs one wavefront: a parity device, not a
// fast path -- the fast path is fga_extend, which keeps genomes resident and runs thousands of units at once.
// ---------------------------------------------------------------------------------------------------
struct shim_work
  { fga_dev *dev;
    uint8_t *dA, *dB; size_t capA, capB;          // packed images on the device
    int4    *pool;    int64_t pool_cells;
    uint16_t *dtrace; int dtrace_cap;
    int     *dout;
    unsigned long long *dcnt;
    std::vector<uint8_t>  pack;
    std::vector<uint16_t> trace;                  // result trace (the reference's work->points)
  };

struct shim_spec
  { double ave_corr; int tspace, reach; float freq[4];
    int path_ave, mscore;
  };

extern "C" void *fga_shim_New_Work_Data(void)
{ shim_work *W = new (std::nothrow) shim_work();
  if (W == NULL) { fga_set_error("out of memory"); return NULL; }
  W->dev = NULL; W->dA = W->dB = NULL; W->capA = W->capB = 0; W->pool = NULL; W->pool_cells = 0;
  W->dtrace = NULL; W->dtrace_cap = 0; W->dout = NULL; W->dcnt = NULL;
  const char *e = getenv("FGA_DEVICE");
  if (fga_dev_open(e != NULL ? atoi(e) : 0,&W->dev))
    { delete W; return NULL; }
  if (hipMalloc(&W->dout,sizeof(int)*8) != hipSuccess || hipMalloc(&W->dcnt,sizeof(unsigned long long)*32) != hipSuccess)
    { fga_set_error("fga_shim_New_Work_Data: device allocation failed");
      hipFree(W->dout); fga_dev_close(W->dev); delete W;
      return NULL;
    }
  return W;
}

extern "C" void fga_shim_Free_Work_Data(void *work)
{ shim_work *W = (shim_work *) work;
  if (W == NULL) return;
  hipSetDevice(W->dev->device);
  hipFree(W->dA); hipFree(W->dB); hipFree(W->pool); hipFree(W->dtrace); hipFree(W->dout); hipFree(W->dcnt);
  fga_dev_close(W->dev);
  delete W;
}

extern "C" void *fga_shim_New_Align_Spec(double ave_corr, int trace_space, float *freq, int reach)
{ shim_spec *S = (shim_spec *) calloc(1,sizeof(shim_spec));
  if (S == NULL) { fga_set_error("out of memory"); return NULL; }
  std::vector<int16_t> tab(2*32768);
  S->ave_corr = ave_corr; S->tspace = trace_space; S->reach = reach;
  for (int k = 0; k < 4; k++) S->freq[k] = freq[k];
  if (fga_align_spec(ave_corr,trace_space,freq,&S->path_ave,tab.data(),tab.data()+32768))
    { free(S); return NULL; }
  S->mscore = tab[32768 + 0x7fff] / 15;
  return S;
}

extern "C" void fga_shim_Free_Align_Spec(void *spec) { free(spec); }

// NUMERIC bytes (0..3) -> the .bps packing (base i in bits 2(i&3) of byte i>>2), zero padding either side
static int shim_upload(shim_work *W, const char *seq, int len, uint8_t **dbuf, size_t *cap)
{ const size_t bytes = (size_t) ((len+3) >> 2) + 2*IMG_PAD + 16;
  W->pack.assign(bytes,0);
  uint8_t *p = W->pack.data() + IMG_PAD;
  for (int i = 0; i < len; i++)
    p[i >> 2] |= (uint8_t) ((seq[i] & 3) << (2*(i & 3)));
  if (*cap < bytes)
    { hipFree(*dbuf); *dbuf = NULL; *cap = 0;
      if (hipMalloc(dbuf,bytes + bytes/4) != hipSuccess)
        { fga_set_error("fga_shim_Local_Alignment: device allocation failed");
          return 1;
        }
      *cap = bytes + bytes/4;
    }
  FGA_HIP(hipMemcpy(*dbuf,W->pack.data(),bytes,hipMemcpyHostToDevice));
  return 0;
}

typedef struct { void *trace; int tlen, diffs, abpos, bbpos, aepos, bepos; } shim_path;            // = Path, align.h:89-95
typedef struct { shim_path *path; uint32_t flags; char *aseq, *bseq; int alen, blen; } shim_alignment;   // = Alignment, 145-152

extern "C" int fga_shim_Local_Alignment(void *align_, void *work, void *spec_, int low, int hgh, int anti, int lbord, int hbord)
{ shim_alignment *align = (shim_alignment *) align_;
  shim_work *W = (shim_work *) work;
  shim_spec *S = (shim_spec *) spec_;
  if (align == NULL || W == NULL || S == NULL || align->path == NULL)
    { fga_set_error("fga_shim_Local_Alignment: null argument");
      return 1;
    }
  if (S->tspace != TS || S->reach != 0)
    { fga_set_error("fga_shim_Local_Alignment: only trace spacing 100 and reach 0 (what FastGA uses, FastGA.c:46, 3757)");
      return 1;
    }
  FGA_HIP(hipSetDevice(W->dev->device));
  const int selfie = (align->aseq == align->bseq);
  if (shim_upload(W,align->aseq,align->alen,&W->dA,&W->capA)) return 1;
  if (!selfie && shim_upload(W,align->bseq,align->blen,&W->dB,&W->capB)) return 1;
  const int maxlen = align->alen > align->blen ? align->alen : align->blen;
  // pool: up to ~64 cells per 100 bases (wide waves) + the trace scratch
  const int64_t need = 64*((int64_t) maxlen/TS + 64) + (1 << ARENA_L0) + 4096;
  if (W->pool_cells < need)
    { hipFree(W->pool); W->pool = NULL; W->pool_cells = 0;
      if (hipMalloc(&W->pool,sizeof(int4)*((size_t) need + 128)) != hipSuccess)
        { fga_set_error("fga_shim_Local_Alignment: device allocation failed");
          return 1;
        }
      W->pool_cells = need;
    }
  const int tcap = 4*(align->alen/TS + 4) + 16;
  if (W->dtrace_cap < tcap)
    { hipFree(W->dtrace); W->dtrace = NULL; W->dtrace_cap = 0;
      if (hipMalloc(&W->dtrace,sizeof(uint16_t)*(size_t) tcap) != hipSuccess)
        { fga_set_error("fga_shim_Local_Alignment: device allocation failed");
          return 1;
        }
      W->dtrace_cap = tcap;
    }
  la_one_args L;
  memset(&L,0,sizeof(L));
  L.G.imgA = (const uint32_t *) W->dA; L.G.imgAr = L.G.imgA;
  L.G.imgB = (const uint32_t *) (selfie ? W->dA : W->dB);
  L.G.padA = L.G.padB = IMG_PAD;
  L.G.tspace = TS; L.G.path_ave = S->path_ave; L.G.mscore = S->mscore;
  L.G.force_lds = getenv("FGA_EXTEND_FORCE_LDS") != NULL;
  L.G.pool = W->pool; L.G.pool_cells = W->pool_cells; L.G.pool_next = W->dcnt + 16; L.G.counters = W->dcnt;
  L.alen = align->alen; L.blen = selfie ? align->alen : align->blen;
  L.acomp = (align->flags & 0x2) != 0;           // ACOMP_FLAG (align.h:128)
  L.selfie = selfie;
  L.low = low; L.hgh = hgh; L.anti = anti; L.lbord = lbord; L.hbord = hbord;
  L.out = W->dout; L.trace_out = W->dtrace; L.trace_cap = W->dtrace_cap;
  hipMemsetAsync(W->dcnt,0,sizeof(unsigned long long)*32,W->dev->stream);
  hipLaunchKernelGGL(local_alignment_one_kernel,dim3(1),dim3(64),0,W->dev->stream,L);
  int out[8];
  hipError_t e = hipMemcpyAsync(out,W->dout,sizeof(int)*7,hipMemcpyDeviceToHost,W->dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(W->dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("fga_shim_Local_Alignment: kernel failed: %s",hipGetErrorString(e));
      return 1;
    }
  if (out[0] != 0)
    { fga_set_error("fga_shim_Local_Alignment: %s",out[0] == 1 ? "trace-point pool exhausted"
                                                    : out[0] == 2 ? "wave wider than the LDS ring (512 diagonals)"
                                                                  : "trace longer than its buffer");
      return 1;
    }
  shim_path *P = align->path;
  P->abpos = out[1]; P->bbpos = out[2]; P->aepos = out[3]; P->bepos = out[4]; P->diffs = out[5]; P->tlen = out[6];
  W->trace.resize((size_t) (P->tlen > 0 ? P->tlen : 1));
  if (P->tlen > 0)
    FGA_HIP(hipMemcpy(W->trace.data(),W->dtrace,sizeof(uint16_t)*(size_t) P->tlen,hipMemcpyDeviceToHost));
  P->trace = W->trace.data();
  return 0;
}
