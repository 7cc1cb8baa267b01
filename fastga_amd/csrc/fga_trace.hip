// fga_trace.hip -- edit scripts of finished alignments from their trace points, on the device (gfx950).
//
// Replaces Compute_Trace_PTS (reference align.c:6171-6308) as every reader of a .1aln calls it (ALNtoPAF.c:278,
// ALNshow.c:524, ALNtoPSL.c:193, ONEaln.c:1011: mode GREEDIEST, no band) and the O(NP) comparison it runs between
// successive trace points, iter_np (align.c:5584-5903).  An alignment is cut by its trace points into panels of at
// most `tspace` bases of A against at most 255 bases of B, each with a known number of differences; the panels are
// independent, so the unit of work is one panel and one lane solves one panel:
//
//   trace_walk_kernel   wave per alignment: <false> number of raw output slots and of scratch cells its panels need,
//                                           <true> one 40-byte descriptor per panel (prefix scans over the trace points)
//   trace_panel_kernel  lane per panel:     the furthest-reaching wave rows (16-bit cells: reach + move code), pointer
//                                           reversal, forward walk emitting the indels into the panel's raw slots
//   trace_count_kernel  wave per alignment: indels and differences of the alignment
//   fga_scan_counts     (fga_chain.hip)     exclusive prefix of the per-alignment counts
//   trace_pack_kernel   wave per alignment: raw slots -> the dense int stream
//
// Cells: cost row d = -2..D, diagonal k (A index = B index + k); a cell holds reach+2 in its low 12 bits and the
// move that produced it (code+1) above them.  The rows are swept exactly in the reference's order (above the target
// diagonal downwards, below it upwards, then the target), because moves towards the target diagonal are free and
// read the row being written.  A panel's row budget is its own difference count minus |M-N| (the reference allows
// the largest count of the whole alignment; on a consistent trace the optimum never needs more than its own).
#include "fga_device.hpp"
#include "fga_gapcore.inc"

namespace {

struct trace_panel
  { int32_t  aln;
    int32_t  ab, bb;          // start of the panel in the (possibly complemented) contigs
    uint16_t M, N;            // bases of A and B
    uint16_t budget;          // cost rows 0..budget
    uint16_t flags;           // 1: budget < 0, the trace is inconsistent
    int64_t  scr;             // first scratch cell
    int64_t  raw;             // first raw output slot
  };                          // 40 bytes

struct trace_args
  { const fga_aln *alns;
    const uint8_t *tbytes;
    int64_t        naln;
    int            tspace, self;
    const uint32_t *imgA, *imgB, *imgBr;
    const int64_t  *boffA, *boffB;
    int64_t        padA, padB;
    // per alignment
    int64_t       *need;      // [2*naln]: raw slots, scratch cells
    const int64_t *pbase, *rbase, *sbase;
    int32_t       *atlen, *adiffs, *astat;
    int64_t       *toff;
    // per panel
    trace_panel   *panels;
    int32_t       *pcnt, *pdiff;
    uint16_t      *cells;
    int32_t       *raw, *dense;
    int64_t        a0, a1;    // alignment range of this batch
    int64_t        p0, np;    // panel range of this batch
    int64_t        s0;        // scratch base of this batch
  };

__device__ __forceinline__ int panel_count(const fga_aln &a)
{ return a.tlen >= 2 ? a.tlen >> 1 : 1; }

__device__ __forceinline__ void panel_need(int M, int N, int pd, int &budget, int &W, int &rows)
{ const int del = M-N, adel = del < 0 ? -del : del;
  budget = pd-adel;
  const int b = budget < 0 ? 0 : budget;
  W = adel + 2*(b >> 1) + 3;
  rows = b+3;
}

__device__ __forceinline__ int wave_scan_incl(int v, int lane)
{ for (int o = 1; o < 64; o <<= 1)
    { const int u = __shfl_up(v,o);
      if (lane >= o) v += u;
    }
  return v;
}

// One wave per alignment walks its trace points 64 panels at a time: the A side of panel p is closed form (panels
// end on multiples of tspace), the B side, the raw output slots and the scratch cells are running prefix sums.
// PLAN = false: totals per alignment (need[]); PLAN = true: the panel descriptors.
template <bool PLAN>
__global__ void __launch_bounds__(64) trace_walk_kernel(trace_args T)
{ const int64_t i = (PLAN ? T.a0 : 0) + blockIdx.x;
  const int lane = threadIdx.x;
  const fga_aln a = T.alns[i];
  const uint8_t *t = T.tbytes + a.toff;
  const int np = panel_count(a);
  const int a0 = (a.abpos/T.tspace)*T.tspace;
  int     run_b = 0;
  int64_t run_raw = 0, run_cells = 0;
  trace_panel *out = PLAN ? T.panels + (T.pbase[i] - T.p0) : NULL;
  const int64_t raw0 = PLAN ? T.rbase[i] : 0, scr0 = PLAN ? T.sbase[i] - T.s0 : 0;
  for (int p0 = 0; p0 < np; p0 += 64)
    { const int  p = p0+lane;
      const bool valid = p < np, last = p == np-1;
      const int  nb = (valid && !last) ? t[2*p+1] : 0;
      const int  nbi = wave_scan_incl(nb,lane);
      const int  bb = a.bbpos + run_b + (nbi-nb);
      const int  ab = p == 0 ? a.abpos : a0 + p*T.tspace;
      const int  ae = last ? a.aepos : a0 + (p+1)*T.tspace;
      const int  M = ae-ab, N = last ? a.bepos-bb : nb;
      const int  pd = valid ? (a.tlen >= 2 ? t[2*p] : a.diffs) : 0;
      int budget, W, rows;
      panel_need(M,N,pd,budget,W,rows);
      const bool bad = budget < 0 || M < 0 || N < 0 || M > 4000 || N > 4000;
      const int  cells = (valid && !bad) ? W*rows : 0;
      const int  ci = wave_scan_incl(cells,lane), ri = wave_scan_incl(pd,lane);
      if (PLAN && valid)
        { trace_panel P;
          P.aln = (int32_t) i; P.ab = ab; P.bb = bb;
          P.M = (uint16_t) M; P.N = (uint16_t) N;
          P.budget = (uint16_t) (budget < 0 ? 0 : budget);
          P.flags = (uint16_t) (bad ? 1 : 0);
          P.scr = scr0 + run_cells + (ci-cells);
          P.raw = raw0 + run_raw + (ri-pd);
          out[p] = P;
        }
      run_b += __shfl(nbi,63);
      run_cells += __shfl(ci,63);
      run_raw += __shfl(ri,63);
    }
  if (!PLAN && lane == 0)
    { T.need[2*i] = run_raw;
      T.need[2*i+1] = run_cells;
    }
}

// 16 bases starting at 2-bit index `idx` of a packed image, base t in bits 2t
__device__ __forceinline__ uint32_t window16(const uint32_t *img, int64_t idx)
{ const int64_t w = idx >> 4;
  const int s = (int) (idx & 15) << 1;
  const uint64_t two = (uint64_t) img[w] | ((uint64_t) img[w+1] << 32);
  return (uint32_t) (two >> s);
}

#define CELL_REACH(c)  ((int) ((c) & 0xfff) - 2)
#define CELL_MOVE(c)   ((int) ((c) >> 12) - 1)
#define MAKE_CELL(j,e) ((uint16_t) (((j)+2) | (((e)+1) << 12)))

// The forward sweep of one panel.  Every cell goes to the panel's scratch in HBM (the walk back needs the cells of
// its path); with LIVE the three rows the sweep READS (d, d-1, d-2) also live in LDS -- lane-interleaved, cell (row, kk)
// of lane l at [(row*TRACE_WL + kk)*64 + l], which is conflict-free whatever kk the lanes are at -- so the sweep issues
// one 2-byte store per cell to HBM and no load.  Panels wider than TRACE_WL cells sweep on the scratch alone.
// Returns the cost row the target was reached in, or -1 (the trace is inconsistent).
#define TRACE_WL 24

template <bool LIVE>
__device__ __forceinline__ int panel_sweep(const trace_args &T, uint16_t *C, uint16_t *L, const uint32_t *imgB,
                                           int64_t abase, int64_t bbase, int M, int N, int dmax, int kmin, int W,
                                           int posl, int posh)
{ const int del = M-N;
  int r0 = 2*TRACE_WL, r1 = TRACE_WL, r2 = 0;          // LDS rows of d, d-1, d-2 (rotated every row)
#define AT(d,k)      C[((d)+2)*W + ((k)-kmin)]
#define LV(r,k)      L[((r) + ((k)-kmin))*64]
#define RD0(d,k)     (LIVE ? LV(r0,k) : AT(d,k))
#define RD1(d,k)     (LIVE ? LV(r1,k) : AT((d)-1,k))
#define RD2(d,k)     (LIVE ? LV(r2,k) : AT((d)-2,k))
#define WR0(d,k,v)   { const uint16_t v_ = (v); AT(d,k) = v_; if (LIVE) LV(r0,k) = v_; }

  int low = del < 0 ? del : 0, hgh = del < 0 ? 0 : del;
  for (int k = low-1; k <= hgh+1; k++)                 // rows -2 and -1
    { const uint16_t c1 = (k == 0) ? MAKE_CELL(-1,0) : MAKE_CELL(-2,0);
      AT(-2,k) = MAKE_CELL(-2,0); AT(-1,k) = c1;
      if (LIVE) { LV(r2,k) = MAKE_CELL(-2,0); LV(r1,k) = c1; }
    }
  low += 1;
  hgh -= 1;

  int d;
  for (d = 0; ; d++)
    { if (d > dmax)
        return -1;
      if ((d & 1) == 0)
        { if (low > posl) low -= 1;
          if (hgh < posh) hgh += 1;
        }
      WR0(d,hgh+1,MAKE_CELL(-2,0))
      WR0(d,low-1,MAKE_CELL(-2,0))

      int j, e;
#define CHOOSE(k,am,ap,mcode,pcode)                             \
      { const int ac = CELL_REACH(RD1(d,k)) + 1;                \
        if (ac < (am))                                          \
          { if ((ap) < (am)) { e = (mcode); j = (am); }         \
            else             { e = (pcode); j = (ap); }         \
          }                                                     \
        else                                                    \
          { if ((ap) < ac)   { e = 0;       j = ac;   }         \
            else             { e = (pcode); j = (ap); }         \
          }                                                     \
      }
#define SLIDE(k)                                                \
      { const int lim = N < M-(k) ? N : M-(k);                  \
        if (j >= 0 && j < lim)                                  \
          { do                                                  \
              { const uint32_t x = window16(T.imgA,abase+(k)+j) ^ window16(imgB,bbase+j);  \
                if (x != 0)                                     \
                  { j += __builtin_ctz(x) >> 1;                 \
                    break;                                      \
                  }                                             \
                j += 16;                                        \
              }                                                 \
            while (j < lim);                                    \
            if (j > lim) j = lim;                               \
          }                                                     \
        WR0(d,k,MAKE_CELL(j,e))                                 \
      }

      j = -2;
      for (int k = hgh; k > del; k--)
        { const int ap = j+1, am = CELL_REACH(RD2(d,k-1));
          CHOOSE(k,am,ap,-1,4)
          SLIDE(k)
        }
      j = -2;
      for (int k = low; k < del; k++)
        { const int ap = CELL_REACH(RD2(d,k+1)) + 1, am = j;
          CHOOSE(k,am,ap,2,1)
          SLIDE(k)
        }
      { const int ap = CELL_REACH(RD0(d,del+1)) + 1, am = j;
        CHOOSE(del,am,ap,2,4)
        SLIDE(del)
      }
      if (j >= N)
        break;
      { const int t = r2; r2 = r1; r1 = r0; r0 = t; }
    }
  return d;
#undef CHOOSE
#undef SLIDE
#undef AT
#undef LV
#undef RD0
#undef RD1
#undef RD2
#undef WR0
}

__global__ void __launch_bounds__(64) trace_panel_kernel(trace_args T)
{ __shared__ uint16_t live[3*TRACE_WL*64];
  const int64_t q = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= T.np) return;
  const trace_panel P = T.panels[q];
  const int64_t gq = T.p0 + q;
  if (P.flags)
    { T.pcnt[gq] = 0; T.pdiff[gq] = -1;
      return;
    }
  const fga_aln a = T.alns[P.aln];
  const int M = P.M, N = P.N, del = M-N, dmax = P.budget;
  const int kmin = (del < 0 ? del : 0) - (dmax >> 1) - 1;
  const int W = (del < 0 ? -del : del) + 2*(dmax >> 1) + 3;
  uint16_t *C = T.cells + P.scr;
#define AT(d,k) C[((d)+2)*W + ((k)-kmin)]

  const bool comp = (a.flags & 0x1) != 0;
  const uint32_t *imgB = comp ? T.imgBr : T.imgB;
  const int64_t abase = (T.padA + T.boffA[a.aread])*4 + P.ab;
  const int64_t bbase = (T.padB + T.boffB[a.bread])*4 + P.bb;

  int posl = -0x3fffffff, posh = 0x3fffffff;         // a self comparison stays on its side of the main diagonal
  if (T.self && a.aread == a.bread && !comp)
    { const int db = P.ab-P.bb;
      if (a.abpos-a.bbpos < 0) posh = -1-db; else posl = 1-db;
    }

  int d = (W <= TRACE_WL) ? panel_sweep<true>(T,C,live + threadIdx.x,imgB,abase,bbase,M,N,dmax,kmin,W,posl,posh)
                          : panel_sweep<false>(T,C,live,imgB,abase,bbase,M,N,dmax,kmin,W,posl,posh);
  if (d < 0)
    { T.pcnt[gq] = 0; T.pdiff[gq] = -1;
      return;
    }

  // reverse the move list from (d,del) back to (0,0), then walk it forwards emitting the indels
  { int k = del, e, h, m;
    uint16_t c;
    c = AT(0,0); AT(0,0) = (uint16_t) ((c & 0xfff) | (4 << 12));
    c = AT(d,k); e = CELL_MOVE(c); AT(d,k) = (uint16_t) ((c & 0xfff) | (4 << 12));
    if (d == 0 && k == 0) e = 3;
    while (e != 3)
      { h = k+e;
        if (e > 1)       h -= 3;
        else if (e == 0) d -= 1;
        else             d -= 2;
        c = AT(d,h);
        m = CELL_MOVE(c);
        AT(d,h) = (uint16_t) ((c & 0xfff) | ((e+1) << 12));
        e = m;
        k = h;
      }
    int32_t *out = T.raw + P.raw;
    int n = 0;
    k = d = 0;
    e = CELL_MOVE(AT(0,0));
    while (e != 3)
      { const int r = CELL_REACH(AT(d,k));
        h = k-e;
        if (e > 1)       h += 3;
        else if (e == 0) d += 1;
        else             d += 2;
        if (h > k)
          out[n++] = P.bb + 1 + r;
        else if (h < k)
          out[n++] = -(P.ab + 1 + r + k);
        k = h;
        e = CELL_MOVE(AT(d,h));
      }
    T.pcnt[gq] = n;
    T.pdiff[gq] = d + (del < 0 ? -del : del);
  }
#undef AT
}

__device__ __forceinline__ int wave_sum(int v)
{ for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v,o);
  return v;
}

__global__ void __launch_bounds__(64) trace_count_kernel(trace_args T)
{ const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int np = panel_count(T.alns[i]);
  const int64_t pb = T.pbase[i];
  int cnt = 0, dif = 0, bad = 0;
  for (int p = lane; p < np; p += 64)
    { const int pdv = T.pdiff[pb+p];
      cnt += T.pcnt[pb+p];
      if (pdv < 0) bad = 1; else dif += pdv;
    }
  cnt = wave_sum(cnt); dif = wave_sum(dif); bad = wave_sum(bad);
  if (lane == 0)
    { T.atlen[i] = cnt; T.adiffs[i] = dif; T.astat[i] = bad; }
}

__global__ void __launch_bounds__(64) trace_pack_kernel(trace_args T)
{ const int64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const int np = panel_count(T.alns[i]);
  const int64_t pb = T.pbase[i];
  int32_t *dst = T.dense + T.toff[i];
  int base = 0;
  for (int p0 = 0; p0 < np; p0 += 64)
    { const int p = p0+lane;
      const int c = p < np ? T.pcnt[pb+p] : 0;
      int inc = c;
      for (int o = 1; o < 64; o <<= 1)
        { const int v = __shfl_up(inc,o);
          if (lane >= o) inc += v;
        }
      if (p < np)
        { const int32_t *src = T.raw + T.panels[pb+p].raw;
          int32_t *d = dst + base + (inc-c);
          for (int x = 0; x < c; x++)
            d[x] = src[x];
        }
      base += __shfl(inc,63);
    }
}

// Gap_Improver, one lane per alignment (fga_gapcore.inc): persistent wavefronts take 64 alignments at a time from a list
// ordered by script length (longest first, so that the lanes of a wavefront have similar work and the long ones start
// early); a lane's F / H scratch is lane-interleaved in the wavefront's slot.  Alignments with more than `tmax` indels
// are left to the host's formatter threads (resume 0): a lane is a chain of dependent loads, ~100 us per indel against
// the 0.3 us of a host thread -- 131 k of them beat 32 threads tenfold on 10^6 short alignments (57 ms), but one lane
// would hold the launch for a long one (512 indels: 50 ms).
struct regroup_args
  { const fga_aln  *alns;
    const int64_t  *toff;
    int32_t        *dense, *adiffs, *resume;
    const int32_t  *order;
    int64_t         n;
    const uint32_t *imgA, *imgB, *imgBr;
    const int64_t  *boffA, *boffB, *clenA, *clenB;
    int64_t         padA, padB;
    int32_t        *F;
    uint16_t       *H;
    int             fcap, tmax;
    int64_t         hcap;
    unsigned int   *ticket;
  };

__global__ void __launch_bounds__(64) trace_regroup_kernel(regroup_args R)
{ const int lane = threadIdx.x;
  int32_t  *F = R.F + (int64_t) blockIdx.x*R.fcap*64 + lane;
  uint16_t *H = R.H + (int64_t) blockIdx.x*R.hcap*64 + lane;
  for (;;)
    { unsigned int c = 0;
      if (lane == 0) c = atomicAdd(R.ticket,1u);
      c = __shfl(c,0);
      const int64_t k = (int64_t) c*64 + lane;
      if ((int64_t) c*64 >= R.n)
        break;
      if (k >= R.n)
        continue;
      const int64_t i = R.order[k];
      const fga_aln a = R.alns[i];
      const int64_t o = R.toff[i];
      const int T = (int) (R.toff[i+1]-o);
      if (T < 2)
        { R.resume[i] = -1;
          continue;
        }
      if (T > R.tmax)
        { R.resume[i] = 0;
          continue;
        }
      const bool comp = (a.flags & 0x1) != 0;
      gap_seq SA, SB;
      SA.img = R.imgA; SA.z = (R.padA + R.boffA[a.aread])*4 - 1; SA.lo = 1; SA.hi = (int) R.clenA[a.aread];
      SB.img = comp ? R.imgBr : R.imgB; SB.z = (R.padB + R.boffB[a.bread])*4 - 1; SB.lo = a.bbpos+1; SB.hi = a.bepos;
      int gained = 0;
      const int r = gap_regroup_packed<64>(SA,SB,SA.hi,(int) R.clenB[a.bread],a.abpos,a.bbpos,R.dense+o,T,
                                           F,R.fcap,H,R.hcap,&gained);
      R.resume[i] = r;
      if (gained != 0)
        R.adiffs[i] += gained;
    }
}

}  // namespace

extern "C" void fga_traces_free(fga_traces *t)
{ if (t == NULL) return;
  free(t->toff); free(t->tlen); free(t->diffs); free(t->trace); free(t->resume);
  free(t);
}

static int trace_pts_impl(fga_dev *dev, const fga_dgenome *GA, const fga_dgenome *GB, const fga_alns *alns,
                          int tspace, int self, int regroup, fga_traces **out)
{ *out = NULL;
  FGA_HIP(fga_dev_enter(dev));
  fga_traces *R = (fga_traces *) calloc(1,sizeof(fga_traces));
  if (R == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  const int64_t n = alns->naln;
  R->naln = n;
  dev->last_ms[FGA_STAGE_TRACE] = dev->last_ms[FGA_STAGE_REGROUP] = 0.f;
  if (n == 0)
    { *out = R;
      return 0;
    }
  int any_comp = 0;
  int64_t npan = 0;
  std::vector<int64_t> pbase(n+1);
  for (int64_t i = 0; i < n; i++)
    { const fga_aln &a = alns->alns[i];
      if (a.flags & 0x1) any_comp = 1;
      pbase[i] = npan;
      npan += a.tlen >= 2 ? a.tlen >> 1 : 1;
      if ((a.tlen & 1) || a.tlen < 0 || a.toff < 0 || a.toff + a.tlen > alns->ntrace)
        { fga_set_error("fga_trace_pts: alignment %lld: trace of %d bytes at offset %lld does not fit the set's %lld "
                        "trace bytes (or is odd)",(long long) i,a.tlen,(long long) a.toff,(long long) alns->ntrace);
          free(R);
          return 1;
        }
      if (a.aread < 0 || a.aread >= GA->nctg || a.bread < 0 || a.bread >= GB->nctg ||
          a.abpos < 0 || a.aepos < a.abpos || a.bbpos < 0 || a.bepos < a.bbpos ||
          (GA->hclen != NULL && a.aepos > GA->hclen[a.aread]) || (GB->hclen != NULL && a.bepos > GB->hclen[a.bread]))
        { fga_set_error("fga_trace_pts: alignment %lld refers to contigs or positions outside the two genomes",
                        (long long) i);
          free(R);
          return 1;
        }
    }
  pbase[n] = npan;
  if (any_comp && GB->img_rc == NULL)
    { fga_set_error("fga_trace_pts: genome 2 was uploaded without its reverse-complement image");
      free(R);
      return 1;
    }

  int rc = 1;
  hipError_t e = hipSuccess;
  fga_aln *d_alns = NULL; uint8_t *d_tb = NULL;
  int64_t *d_need = NULL, *d_pbase = NULL, *d_rbase = NULL, *d_sbase = NULL, *d_toff = NULL, *d_tsum = NULL;
  int32_t *d_atlen = NULL, *d_adiffs = NULL, *d_astat = NULL, *d_pcnt = NULL, *d_pdiff = NULL;
  int32_t *d_raw = NULL, *d_dense = NULL;
  trace_panel *d_panels = NULL; uint16_t *d_cells = NULL;
  int32_t *d_order = NULL, *d_resume = NULL, *d_F = NULL; uint16_t *d_H = NULL; unsigned int *d_ticket = NULL;
  hipEvent_t evr0 = NULL, evr1 = NULL;
  std::vector<int64_t> need(2*n), rbase(n+1), sbase(n+1);
  trace_args T;
  memset(&T,0,sizeof(T));
  int64_t total = 0, rawtot = 0;
  int64_t cell_cap = (int64_t) 4 << 30;                // scratch cells per batch (8 GB)
  { const char *ev = getenv("FGA_TRACE_CELLS");        // test hook: small batches
    if (ev != NULL && atoll(ev) > 0) cell_cap = atoll(ev);
  }
  const size_t tbbytes = (size_t) (alns->ntrace > 0 ? alns->ntrace : 1);

#define TRY(call) do { if ((e = (call)) != hipSuccess) goto fail; } while (0)
  TRY(fga_dmalloc(&d_alns,sizeof(fga_aln)*n));
  TRY(fga_dmalloc(&d_tb,tbbytes+16));
  TRY(fga_dmalloc(&d_need,sizeof(int64_t)*2*n));
  TRY(fga_dmalloc(&d_pbase,sizeof(int64_t)*(n+1)));
  TRY(fga_dmalloc(&d_rbase,sizeof(int64_t)*(n+1)));
  TRY(fga_dmalloc(&d_sbase,sizeof(int64_t)*(n+1)));
  TRY(fga_dmalloc(&d_toff,sizeof(int64_t)*(n+1)));
  TRY(fga_dmalloc(&d_atlen,sizeof(int32_t)*(n+1)));                    // one more, zero: the scan's last offset is the total
  TRY(fga_dmalloc(&d_tsum,sizeof(int64_t)*((n+1+4095)/4096 + 1)));
  TRY(fga_dmalloc(&d_adiffs,sizeof(int32_t)*n));
  TRY(fga_dmalloc(&d_astat,sizeof(int32_t)*n));
  TRY(fga_dmalloc(&d_pcnt,sizeof(int32_t)*npan));
  TRY(fga_dmalloc(&d_pdiff,sizeof(int32_t)*npan));
  TRY(hipMemcpyAsync(d_alns,alns->alns,sizeof(fga_aln)*n,hipMemcpyHostToDevice,dev->stream));
  if (alns->ntrace > 0)
    TRY(hipMemcpyAsync(d_tb,alns->tbytes,alns->ntrace,hipMemcpyHostToDevice,dev->stream));
  TRY(hipMemcpyAsync(d_pbase,pbase.data(),sizeof(int64_t)*(n+1),hipMemcpyHostToDevice,dev->stream));

  T.alns = d_alns; T.tbytes = d_tb; T.naln = n; T.tspace = tspace; T.self = self;
  T.imgA = (const uint32_t *) GA->img; T.imgB = (const uint32_t *) GB->img; T.imgBr = (const uint32_t *) GB->img_rc;
  T.boffA = GA->boff; T.boffB = GB->boff; T.padA = GA->pad; T.padB = GB->pad;
  T.need = d_need; T.pbase = d_pbase; T.rbase = d_rbase; T.sbase = d_sbase; T.toff = d_toff;
  T.atlen = d_atlen; T.adiffs = d_adiffs; T.astat = d_astat; T.pcnt = d_pcnt; T.pdiff = d_pdiff;

  hipEventRecord(dev->ev0,dev->stream);
  hipLaunchKernelGGL(trace_walk_kernel<false>,dim3((unsigned) n),dim3(64),0,dev->stream,T);
  TRY(hipMemcpyAsync(need.data(),d_need,sizeof(int64_t)*2*n,hipMemcpyDeviceToHost,dev->stream));
  TRY(hipStreamSynchronize(dev->stream));
  { int64_t r = 0, s = 0;
    for (int64_t i = 0; i < n; i++)
      { rbase[i] = r; sbase[i] = s;
        r += need[2*i]; s += need[2*i+1];
      }
    rbase[n] = r; sbase[n] = s;
    rawtot = r;
  }
  TRY(fga_dmalloc(&d_raw,sizeof(int32_t)*(rawtot+1)));
  TRY(hipMemcpyAsync(d_rbase,rbase.data(),sizeof(int64_t)*(n+1),hipMemcpyHostToDevice,dev->stream));
  TRY(hipMemcpyAsync(d_sbase,sbase.data(),sizeof(int64_t)*(n+1),hipMemcpyHostToDevice,dev->stream));
  T.raw = d_raw;

  { // batches of whole alignments bounded by scratch cells
    int64_t maxcells = 0;
    std::vector<int64_t> cut(1,0);
    for (int64_t i = 0; i < n; )
      { int64_t j = i+1;
        while (j < n && sbase[j+1]-sbase[i] <= cell_cap) j++;
        cut.push_back(j);
        if (sbase[j]-sbase[i] > maxcells) maxcells = sbase[j]-sbase[i];
        i = j;
      }
    TRY(fga_dmalloc(&d_cells,sizeof(uint16_t)*(maxcells+1)));
    TRY(fga_dmalloc(&d_panels,sizeof(trace_panel)*npan));
    T.cells = d_cells;
    for (size_t b = 0; b+1 < cut.size(); b++)
      { T.a0 = cut[b]; T.a1 = cut[b+1];
        T.p0 = pbase[T.a0]; T.np = pbase[T.a1]-T.p0;
        T.s0 = sbase[T.a0];
        T.panels = d_panels + T.p0;
        const int64_t na = T.a1-T.a0;
        hipLaunchKernelGGL(trace_walk_kernel<true>,dim3((unsigned) na),dim3(64),0,dev->stream,T);
        hipLaunchKernelGGL(trace_panel_kernel,dim3((unsigned) ((T.np+63)/64)),dim3(64),0,dev->stream,T);
      }
    T.panels = d_panels; T.p0 = 0;
  }
  hipLaunchKernelGGL(trace_count_kernel,dim3((unsigned) n),dim3(64),0,dev->stream,T);
  TRY(hipMemsetAsync(d_atlen+n,0,sizeof(int32_t),dev->stream));
  fga_scan_counts(dev,d_atlen,n+1,d_tsum,d_toff);
  R->toff = (int64_t *) malloc(sizeof(int64_t)*(n+1));
  R->tlen = (int32_t *) malloc(sizeof(int32_t)*n);
  R->diffs = (int32_t *) malloc(sizeof(int32_t)*n);
  if (R->toff == NULL || R->tlen == NULL || R->diffs == NULL)
    { fga_set_error("out of memory");
      goto done;
    }
  TRY(hipMemcpyAsync(R->toff,d_toff,sizeof(int64_t)*(n+1),hipMemcpyDeviceToHost,dev->stream));
  TRY(hipStreamSynchronize(dev->stream));
  total = R->toff[n];
  TRY(fga_dmalloc(&d_dense,sizeof(int32_t)*(total+1)));
  T.dense = d_dense;
  hipLaunchKernelGGL(trace_pack_kernel,dim3((unsigned) n),dim3(64),0,dev->stream,T);
  hipEventRecord(dev->ev1,dev->stream);
  if (regroup)
    { regroup_args G;
      memset(&G,0,sizeof(G));
      G.fcap = 520; G.hcap = 16384; G.tmax = 512;
      { // What the host's formatter threads would need for all of it (0.3 us per indel and thread) decides what the device
        // takes: a lane needs 35-100 us per indel, so the launch lasts as long as its longest script -- it may cost a
        // quarter of the host's estimate, and is not made at all when that leaves it only scripts of a few indels (the 2 k
        // contig-long alignments of the bench pair: 3 M indels, 9 ms on 32 threads, against 17 ms for 512-indel lanes).
        const int nthr = dev->host_threads > 0 ? dev->host_threads : 8;
        const double host_ms = 0.3e-3 * (double) total / nthr;
        if (host_ms*5. < G.tmax)
          G.tmax = (int) (host_ms*5.);
      }
      if (G.tmax < 64)
        { G.tmax = 1; G.fcap = 2; G.hcap = 2; }             // everything to the host (T < 2 needs no regrouping)
      { const char *ev = getenv("FGA_REGROUP_CAPS");        // test hook / tuning: "<fcap>,<hcap>,<tmax>"
        long long f, h, m;
        if (ev != NULL && sscanf(ev,"%lld,%lld,%lld",&f,&h,&m) == 3 && f > 0 && h > 0 && m > 0)
          { G.fcap = (int) (f > 65535 ? 65535 : f); G.hcap = h > (1 << 20) ? (1 << 20) : h; G.tmax = (int) (m > 65535 ? 65535 : m); }
      }
      if (G.fcap > G.tmax+8)
        G.fcap = G.tmax+8;
      // longest scripts first (counting sort on the length, everything beyond tmax in the first bucket)
      std::vector<int32_t> order((size_t) n);
      { std::vector<int64_t> cnt((size_t) G.tmax+3,0);
        for (int64_t i = 0; i < n; i++)
          { const int64_t l = R->toff[i+1]-R->toff[i];
            cnt[(size_t) (l > G.tmax ? 0 : G.tmax+1-l) + 1] += 1;
          }
        for (int b = 0; b <= G.tmax+1; b++)
          cnt[(size_t) b+1] += cnt[(size_t) b];
        for (int64_t i = 0; i < n; i++)
          { const int64_t l = R->toff[i+1]-R->toff[i];
            order[(size_t) cnt[(size_t) (l > G.tmax ? 0 : G.tmax+1-l)]++] = (int32_t) i;
          }
      }
      int64_t slots = (n+63)/64;
      if (slots > (int64_t) dev->ncu*8) slots = (int64_t) dev->ncu*8;
      TRY(fga_dmalloc(&d_order,sizeof(int32_t)*n));
      TRY(fga_dmalloc(&d_resume,sizeof(int32_t)*n));
      TRY(fga_dmalloc(&d_F,sizeof(int32_t)*64*(size_t) G.fcap*slots));
      TRY(fga_dmalloc(&d_H,sizeof(uint16_t)*64*(size_t) G.hcap*slots));
      TRY(fga_dmalloc(&d_ticket,sizeof(unsigned int)));
      TRY(hipMemcpyAsync(d_order,order.data(),sizeof(int32_t)*n,hipMemcpyHostToDevice,dev->stream));
      TRY(hipMemsetAsync(d_ticket,0,sizeof(unsigned int),dev->stream));
      G.alns = d_alns; G.toff = d_toff; G.dense = d_dense; G.adiffs = d_adiffs; G.resume = d_resume; G.order = d_order;
      G.n = n; G.imgA = T.imgA; G.imgB = T.imgB; G.imgBr = T.imgBr;
      G.boffA = GA->boff; G.boffB = GB->boff; G.clenA = GA->clen; G.clenB = GB->clen; G.padA = GA->pad; G.padB = GB->pad;
      G.F = d_F; G.H = d_H; G.ticket = d_ticket;
      TRY(hipEventCreate(&evr0)); TRY(hipEventCreate(&evr1));
      hipEventRecord(evr0,dev->stream);
      hipLaunchKernelGGL(trace_regroup_kernel,dim3((unsigned) slots),dim3(64),0,dev->stream,G);
      hipEventRecord(evr1,dev->stream);
      R->resume = (int32_t *) malloc(sizeof(int32_t)*n);
      if (R->resume == NULL)
        { fga_set_error("out of memory");
          goto done;
        }
      TRY(hipMemcpyAsync(R->resume,d_resume,sizeof(int32_t)*n,hipMemcpyDeviceToHost,dev->stream));
    }
  R->trace = (int32_t *) malloc(sizeof(int32_t)*(total+1));
  if (R->trace == NULL)
    { fga_set_error("out of memory");
      goto done;
    }
  TRY(hipMemcpyAsync(R->trace,d_dense,sizeof(int32_t)*total,hipMemcpyDeviceToHost,dev->stream));
  TRY(hipMemcpyAsync(R->tlen,d_atlen,sizeof(int32_t)*n,hipMemcpyDeviceToHost,dev->stream));
  TRY(hipMemcpyAsync(R->diffs,d_adiffs,sizeof(int32_t)*n,hipMemcpyDeviceToHost,dev->stream));
  { std::vector<int32_t> stat(n);
    TRY(hipMemcpyAsync(stat.data(),d_astat,sizeof(int32_t)*n,hipMemcpyDeviceToHost,dev->stream));
    TRY(hipStreamSynchronize(dev->stream));
    TRY(hipGetLastError());
    hipEventElapsedTime(&dev->last_ms[FGA_STAGE_TRACE],dev->ev0,dev->ev1);
    if (regroup)
      { hipEventElapsedTime(&dev->last_ms[FGA_STAGE_REGROUP],evr0,evr1);
        if (getenv("FGA_TRACE_TIMING") != NULL)
          { int64_t back = 0;
            for (int64_t i = 0; i < n; i++)
              back += R->resume[i] >= 0;
            fprintf(stderr,"  fga_trace_pts_regrouped: scripts %.1f ms, regrouping %.1f ms (%lld of %lld alignments handed "
                           "back to the host)\n",dev->last_ms[FGA_STAGE_TRACE],dev->last_ms[FGA_STAGE_REGROUP],
                    (long long) back,(long long) n);
          }
      }
    for (int64_t i = 0; i < n; i++)
      if (stat[i] != 0)
        { fga_set_error("fga_trace_pts: alignment %lld: trace points are inconsistent with the sequences "
                        "(Compute_Trace_PTS would fail)",(long long) i);
          goto done;
        }
  }
  R->ntrace = total;
  R->npanels = npan;
  rc = 0;
  goto done;
#undef TRY

fail:
  fga_set_error("fga_trace_pts: %s",hipGetErrorString(e));
done:
  fga_pool_free(d_alns); fga_pool_free(d_tb); fga_pool_free(d_need); fga_pool_free(d_pbase); fga_pool_free(d_rbase); fga_pool_free(d_sbase);
  fga_pool_free(d_toff); fga_pool_free(d_tsum); fga_pool_free(d_atlen); fga_pool_free(d_adiffs); fga_pool_free(d_astat); fga_pool_free(d_pcnt); fga_pool_free(d_pdiff);
  fga_pool_free(d_raw); fga_pool_free(d_dense); fga_pool_free(d_panels); fga_pool_free(d_cells);
  fga_pool_free(d_order); fga_pool_free(d_resume); fga_pool_free(d_F); fga_pool_free(d_H); fga_pool_free(d_ticket);
  if (evr0 != NULL) hipEventDestroy(evr0);
  if (evr1 != NULL) hipEventDestroy(evr1);
  if (rc != 0)
    { fga_traces_free(R);
      return 1;
    }
  *out = R;
  return 0;
}

extern "C" int fga_trace_pts(fga_dev *dev, const fga_dgenome *GA, const fga_dgenome *GB, const fga_alns *alns,
                             int tspace, int self, fga_traces **out)
{ return trace_pts_impl(dev,GA,GB,alns,tspace,self,0,out); }

extern "C" int fga_trace_pts_regrouped(fga_dev *dev, const fga_dgenome *GA, const fga_dgenome *GB, const fga_alns *alns,
                                       int tspace, int self, fga_traces **out)
{ return trace_pts_impl(dev,GA,GB,alns,tspace,self,1,out); }

// ---------------------------------------------------------------------------------------------------------------------
// The host instantiation of fga_gapcore.inc: the check of the device kernel's source where there is no GPU.  The images
// are laid out as fga_dgenome_upload / revcomp_kernel lay them out on the device (fga_extend.hip).
// ---------------------------------------------------------------------------------------------------------------------
#define CHECK_PAD 4096

static uint8_t *host_image(const fga_gdb *G, int rc)
{ const size_t bytes = (size_t) G->bpslen + 2*CHECK_PAD + 16;
  uint8_t *img = (uint8_t *) calloc(bytes,1);
  if (img == NULL)
    return NULL;
  if (!rc)
    { memcpy(img+CHECK_PAD,G->bps,G->bpslen);
      return img;
    }
  for (int c = 0; c < G->ncontig; c++)
    { const int64_t len = G->contigs[c].clen;
      const uint8_t *src = G->bps + G->contigs[c].boff;
      uint8_t *dst = img + CHECK_PAD + G->contigs[c].boff;
      for (int64_t x = 0; x < len; x++)
        { const int64_t y = len-1-x;
          const int b = (src[y >> 2] >> (2*(y & 3))) & 3;
          dst[x >> 2] |= (uint8_t) ((3-b) << (2*(x & 3)));
        }
    }
  return img;
}

extern "C" int fga_gap_core_check(const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns, fga_traces *traces,
                                  int fcap, int64_t hcap)
{ if (g2 == NULL) g2 = g1;
  if (traces == NULL || traces->naln != alns->naln || fcap < 2 || hcap < 2)
    { fga_set_error("fga_gap_core_check: the edit scripts do not belong to this alignment set");
      return 1;
    }
  uint8_t *ia = host_image(g1,0), *ib = host_image(g2,0), *ir = host_image(g2,1);
  int32_t *F = (int32_t *) malloc(sizeof(int32_t)*(size_t) fcap);
  uint16_t *H = (uint16_t *) malloc(sizeof(uint16_t)*(size_t) hcap);
  int rc = 1;
  free(traces->resume);
  traces->resume = (int32_t *) malloc(sizeof(int32_t)*(size_t) (alns->naln > 0 ? alns->naln : 1));
  if (ia == NULL || ib == NULL || ir == NULL || F == NULL || H == NULL || traces->resume == NULL)
    { fga_set_error("fga_gap_core_check: out of memory");
      goto done;
    }
  for (int64_t i = 0; i < alns->naln; i++)
    { const fga_aln &a = alns->alns[i];
      if (a.aread < 0 || a.aread >= g1->ncontig || a.bread < 0 || a.bread >= g2->ncontig || a.abpos < 0 || a.bbpos < 0 ||
          a.aepos > g1->contigs[a.aread].clen || a.bepos > g2->contigs[a.bread].clen)
        { fga_set_error("fga_gap_core_check: alignment %lld lies outside the contigs",(long long) i);
          goto done;
        }
      const bool comp = (a.flags & 0x1) != 0;
      gap_seq SA, SB;
      SA.img = (const uint32_t *) ia; SA.z = (CHECK_PAD + g1->contigs[a.aread].boff)*4 - 1;
      SA.lo = 1; SA.hi = (int) g1->contigs[a.aread].clen;
      SB.img = (const uint32_t *) (comp ? ir : ib); SB.z = (CHECK_PAD + g2->contigs[a.bread].boff)*4 - 1;
      SB.lo = a.bbpos+1; SB.hi = a.bepos;
      int gained = 0;
      traces->resume[i] = traces->tlen[i] < 2 ? -1 :
          gap_regroup_packed<1>(SA,SB,SA.hi,(int) g2->contigs[a.bread].clen,a.abpos,a.bbpos,
                                traces->trace+traces->toff[i],traces->tlen[i],F,fcap,H,hcap,&gained);
      traces->diffs[i] += gained;
    }
  rc = 0;
done:
  free(ia); free(ib); free(ir); free(F); free(H);
  return rc;
}
