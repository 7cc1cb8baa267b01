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
#define ARENA_KEEP  4                   // levels 0..3 (245 k cells, 3.9 MB) stay with the wavefront; larger ones are returned
#define ARENA_LIST_CAP 1024             // returned levels remembered per size
#define ARENA_LEVEL(i)  (31 - __builtin_clz((((unsigned) (i)) >> ARENA_L0) + 1u))
#define ARENA_START(l)  ((int) (((1u << (l)) - 1u) << ARENA_L0))
#define ARENA_END(l)    ((int) (((2u << (l)) - 1u) << ARENA_L0))       // level 15: 2^30 - 2^14, fits an int

struct ext_prof
  { unsigned long long t_steps, t_unwind, t_total, nsteps;
    unsigned long long ncells;        // diagonal updates: sum over wave steps of the wave width (cell updates)
    unsigned long long nbases;        // bases compared by the snakes (each counts once per sequence in B_ext)
  };

struct ext_state              // wave-uniform alignment state (the reference's Path + trace pointer)
  { int abpos, bbpos, aepos, bepos, diffs, tlen;
    int tpos;                 // index of trace[0] inside the per-wave trace scratch
  };

struct arena_lists            // returned arena levels by size, each list behind its own spin lock (fga_extend_kernel.inc)
  { int       lock[ARENA_NLEV];
    int       nfree[ARENA_NLEV];
    long long cell[ARENA_NLEV][ARENA_LIST_CAP];
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
    int            *wide_q;              // narrow kernel: units handed to the full kernel
    // alignment parameters
    int   tspace, path_ave, self, aln_min, mscore, force_lds;
    double aln_rate;
    const int *trim32;                   // the reference's TABLE[32768] (align.c:207-218) as dwords: the scalar-side trim test
    // scratch (per workgroup)
    int4     *pool;   int64_t pool_cells;    // the launch's cell pool (pebble levels and trace scratch of all wavefronts)
    unsigned long long *pool_next;           // its head
    arena_lists *lists;                      // levels returned by finished units
    // output
    fga_aln  *alns; int64_t aln_cap;
    uint8_t  *tbytes; int64_t tbytes_cap;
    unsigned long long *counters;            // [0] alignments, [1] trace bytes, [2] calls, [3] waves, [4] error flag
    unsigned long long *ctg_waves;           // [A contigs, index order] wave steps of the contig's units (NULL: not counted)
  };

struct la_one_args
  { ext_args G;
    int alen, blen, acomp, selfie, low, hgh, anti, lbord, hbord;
    int *out;                  // status, abpos, bbpos, aepos, bepos, diffs, tlen
    uint16_t *trace_out; int trace_cap;
  };

// The device code is compiled twice (fga_extend_kernel.inc):
//   ext_full  512-diagonal ring, 8192-base windows: 27 KB of LDS per wavefront, six wavefronts per CU.  The latency regime
//             (few long units: the run time is the longest unit's serial chain) and the refuge of units handed over by
//   ext_mid   256-diagonal ring in ONE copy that a wave step updates in place, 3072-base windows, the register budget of
//             four wavefronts per SIMD: 8.2 KB of LDS, sixteen resident wavefronts per CU -- the throughput regime
//             (10^5 .. 10^6 short units; no wave of the 150-Mbp repeat-heavy self comparison, the 2 % or the 10 % pair
//             is wider than 248 diagonals).
#define RC  512
#define WDW 512
#define EXT_NS ext_full
#define EXT_KERNEL_ATTR
#define EXT_HANDOFF 0
#define RING_INPLACE 0
#include "fga_extend_kernel.inc"
#undef RC
#undef WDW
#undef EXT_NS
#undef EXT_KERNEL_ATTR
#undef EXT_HANDOFF
#undef RING_INPLACE
#undef RING_NB
#undef RB

#ifndef EXT_MID_RC
#define EXT_MID_RC  256
#define EXT_MID_WDW 192
#define EXT_MID_OCC 5                  // wavefronts per SIMD the register budget is held to (96 VGPRs)
#define EXT_MID_WGS 20                 // resident wavefronts per CU (7.7 KB of LDS each)
#endif
#ifndef EXT_MID_INPLACE
#define EXT_MID_INPLACE 1              // one copy of the ring, updated in place (fga_extend_kernel.inc)
#endif
#define RC  EXT_MID_RC
#define WDW EXT_MID_WDW
#define EXT_NS ext_mid
#define EXT_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(EXT_MID_OCC,EXT_MID_OCC)))
#define EXT_HANDOFF 1
#define RING_INPLACE EXT_MID_INPLACE
#include "fga_extend_kernel.inc"
#undef RC
#undef WDW
#undef EXT_NS
#undef EXT_KERNEL_ATTR
#undef EXT_HANDOFF
#undef RING_INPLACE
#undef RING_NB
#undef RB

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
// the records of a launch in discovery order -- (unit, sequence number inside the unit), what the redundancy filter
// works in -- before they leave the device: a key per record, the LSD sort of fga_sort.hip, a gather of the 56-byte
// records (the trace bytes stay where they are).  The host filter ordered 2.3 M records of a 3 Gbp part in 40-130 ms.
// ---------------------------------------------------------------------------------------------------
__global__ void aln_key_kernel(const fga_aln *alns, int64_t n, uint4 *rec)
{ const int64_t i = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n)
    rec[i] = make_uint4((uint32_t) i,(uint32_t) alns[i].seq,(uint32_t) alns[i].unit,0u);     // sorted on bits 32..: (unit, seq)
}

__global__ void aln_gather_kernel(const uint4 *rec, const fga_aln *alns, int64_t n, fga_aln *out)
{ const int64_t j = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (j < n)
    out[j] = alns[rec[j].x];
}

// units by decreasing estimated work (sum of their hit boxes' lengths), ties by unit index: the order the persistent
// wavefronts take them in.  A key per unit, the LSD sort of fga_sort.hip, the indices out.  (The host team needed 15-20 ms
// for the 2 M units of a 3 Gbp part.)
#define WORK_BITS 40
__global__ void unit_work_kernel(const fga_unit *units, const fga_hit *hits, int64_t nu, uint4 *rec)
{ const int64_t u = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (u >= nu) return;
  int64_t s = 0;
  const fga_hit *h = hits + units[u].first_hit;
  for (int q = 0; q < units[u].nhits; q++)
    s += (h[q].ahgh - h[q].alow) + 1000;
  if (s < 0) s = 0;
  if (s > ((int64_t) 1 << WORK_BITS) - 1) s = ((int64_t) 1 << WORK_BITS) - 1;
  const uint64_t k = (uint64_t) (((int64_t) 1 << WORK_BITS) - 1 - s);         // ascending key = decreasing work
  rec[u] = make_uint4((uint32_t) u,(uint32_t) k,(uint32_t) (k >> 32),0u);
}

__global__ void unit_order_kernel(const uint4 *rec, int64_t nu, int *order)
{ const int64_t j = (int64_t) blockIdx.x*blockDim.x + threadIdx.x;
  if (j < nu)
    order[j] = (int) rec[j].x;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#define IMG_PAD FGA_IMG_PAD

// adopt != NULL: a padded image of G's bases that is on the device already (the index build's, fga_dgix_build_keep) is
// taken over instead of uploading the bases a second time
static int dgenome_make(fga_dev *dev, const fga_gdb *G, const int *perm, int nperm, int want_revcomp, uint8_t *adopt,
                        fga_dgenome **out)
{ *out = NULL;
  FGA_HIP(fga_dev_enter(dev));
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
  if (adopt != NULL)
    D->img = adopt;
  if ((adopt == NULL && (e = fga_dmalloc(&D->img,bytes)) != hipSuccess) ||
      (e = fga_dmalloc(&D->boff,sizeof(int64_t)*G->ncontig)) != hipSuccess ||
      (e = fga_dmalloc(&D->clen,sizeof(int64_t)*G->ncontig)) != hipSuccess ||
      (e = fga_dmalloc(&D->perm,sizeof(int)*(nperm > 0 ? nperm : 1))) != hipSuccess ||
      (want_revcomp && (e = fga_dmalloc(&D->img_rc,bytes)) != hipSuccess))
    { fga_set_error("fga_dgenome_upload: device allocation failed: %s",hipGetErrorString(e));
      fga_pool_free(D->img); fga_pool_free(D->boff); fga_pool_free(D->clen); fga_pool_free(D->perm); fga_pool_free(D->img_rc); free(D);
      return 1;
    }
  // every fill and the kernel below on the context's stream, in order: a hipMemset (legacy default stream, returns before it
  // has happened) is not ordered with a non-blocking stream -- with several contexts opening at once in one process (fga_run_multi)
  // the zeroing of the complement image could land AFTER revcomp_kernel had written it
  if (adopt == NULL)
    { hipMemsetAsync(D->img,0,IMG_PAD,dev->stream);
      hipMemsetAsync(D->img + IMG_PAD + G->bpslen,0,bytes - IMG_PAD - (size_t) G->bpslen,dev->stream);
      hipMemcpyAsync(D->img + IMG_PAD,G->bps,G->bpslen,hipMemcpyHostToDevice,dev->stream);
    }
  hipMemcpy(D->boff,boff.data(),sizeof(int64_t)*G->ncontig,hipMemcpyHostToDevice);
  hipMemcpy(D->clen,clen.data(),sizeof(int64_t)*G->ncontig,hipMemcpyHostToDevice);
  D->hclen = (int64_t *) malloc(sizeof(int64_t)*(G->ncontig > 0 ? G->ncontig : 1));
  if (D->hclen != NULL)
    memcpy(D->hclen,clen.data(),sizeof(int64_t)*G->ncontig);
  if (nperm > 0)
    hipMemcpy(D->perm,perm,sizeof(int)*nperm,hipMemcpyHostToDevice);
  if (want_revcomp)
    { hipMemsetAsync(D->img_rc,0,bytes,dev->stream);
      dim3 grid(256,G->ncontig);
      hipLaunchKernelGGL(revcomp_kernel,grid,dim3(256),0,dev->stream,D->img,D->img_rc,D->boff,D->clen,
                         G->ncontig,(int64_t) IMG_PAD);
    }
  e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("fga_dgenome_upload: %s",hipGetErrorString(e));
      fga_pool_free(D->img); fga_pool_free(D->boff); fga_pool_free(D->clen); fga_pool_free(D->perm); fga_pool_free(D->img_rc); free(D->hclen); free(D);
      return 1;
    }
  *out = D;
  return 0;
}

extern "C" int fga_dgenome_upload(fga_dev *dev, const fga_gdb *G, const int *perm, int nperm, int want_revcomp,
                                  fga_dgenome **out)
{ return dgenome_make(dev,G,perm,nperm,want_revcomp,NULL,out); }

extern "C" int fga_dgenome_adopt(fga_dev *dev, const fga_gdb *G, const int *perm, int nperm, int want_revcomp,
                                 void *image, fga_dgenome **out)
{ if (image == NULL)
    { *out = NULL;
      fga_set_error("fga_dgenome_adopt: no image");
      return 1;
    }
  return dgenome_make(dev,G,perm,nperm,want_revcomp,(uint8_t *) image,out);
}

extern "C" void fga_dgenome_free(fga_dgenome *D)
{ if (D == NULL) return;
  fga_dev_enter(D->dev);
  fga_pool_free(D->img); fga_pool_free(D->img_rc); fga_pool_free(D->boff); fga_pool_free(D->clen); fga_pool_free(D->perm);
  free(D->hclen);
  free(D);
}

extern "C" int fga_extend(fga_dev *dev, const fga_dgenome *GA, const fga_dgenome *GB, const fga_hits *H,
                          const fga_extend_params *prm, fga_alns **out)
{ *out = NULL;
  FGA_HIP(fga_dev_enter(dev));
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

  // resident single-wavefront workgroups: the LDS block of a wavefront (26.5 KB) allows six per CU.  With few units the
  // run time is the longest unit's serial chain, which is fastest with four (it shares its CU's issue slots, LDS and L1
  // with fewer neighbours: 80.7 vs 83.9 ms on the bench pair); with many units throughput wins (1 M units: 792 -> 647 ms)
  // Throughput regime (>= 16 units per wavefront slot): the ext_mid build (eleven resident wavefronts per CU) over all
  // units, then -- if any wave outgrew its 256-diagonal ring -- the ext_full build over the units handed over.
  // FGA_EXTEND_NARROW=0 / 1 overrides the choice.
  const bool many = H->nunits >= 16*(int64_t) dev->ncu*4;
  bool narrow = many;
  { const char *ev = getenv("FGA_EXTEND_NARROW");
    if (ev != NULL) narrow = atoi(ev) != 0;
    if (getenv("FGA_EXTEND_FORCE_LDS") != NULL) narrow = false;
  }
  // Scratch and output are sized by what the hits suggest, not by the worst case times the number of wavefronts; the
  // kernel counts what it would have needed, so a launch that runs out is repeated once with exactly that
  // (prm->cell_cap: pool cells; prm->aln_cap / prm->trace_cap: output records / trace bytes).
  int64_t span = 0, span_max = 0;              // hit-box bases of all units / of the unit with the most
  for (int64_t u = 0; u < H->nunits; u++)
    { int64_t su = 0;
      for (int64_t h = H->units[u].first_hit, e = h + H->units[u].nhits; h < e; h++)
        su += (H->hits[h].ahgh - H->hits[h].alow) / 2 + 200;
      span += su;
      if (su > span_max) span_max = su;
    }
  // Latency regime: the run time is the longest unit's serial chain, and that chain is shortest when its wavefront has the CU
  // to itself (bench pair, 20 launches each: four wavefronts per CU 58.1 .. 69.0 ms, mean 60.5; three 59.6; two 59.0; ONE 56.4 ..
  // 58.5, mean 56.8 -- the neighbours share the CU's scalar unit, caches and LDS).  So: as few wavefronts per CU as leave
  // every wavefront's share of the work below half of the longest unit's (both as hit-box bases), four at most.
  int per_cu = 4;
  if (!many)
    for (per_cu = 1; per_cu < 4; per_cu++)
      if (2*span <= span_max * (int64_t) dev->ncu * per_cu)
        break;
  int nwg = dev->ncu * (narrow ? EXT_MID_WGS : (many ? 6 : per_cu));
  { const char *ev = getenv("FGA_EXTEND_WGS");
    if (ev != NULL && atoi(ev) > 0) nwg = atoi(ev);
  }
  if (nwg > H->nunits) nwg = (int) H->nunits;
  // cells / trace bytes per hit-box base: 0.12 / 0.04 to begin with, afterwards what the device context's last launch needed
  // plus a quarter.  (Measured beyond every wavefront's first level: 0.02-0.03 cells per base -- 3 Gbp at 10 %: 4.0 GB used,
  // 150 Mbp repeat-heavy self: 1.1 GB, the 100 Mbp bench pair: 0.33 GB; the first estimate, 0.48, asked for 63.6 / 5.7 /
  // 2.6 GB, and at 3 Gbp the allocation of those 64 GB cost 0.6-3.2 s.  A launch that does run out is repeated once with
  // what it counted.)
  const double cpb = dev->ext_cells_per_base > 0. ? 1.25*dev->ext_cells_per_base : 0.12;
  const double tpb = dev->ext_tbytes_per_base > 0. ? 1.25*dev->ext_tbytes_per_base : 0.04;
  int64_t pool_cells = prm->cell_cap > 0 ? prm->cell_cap
                                         : (int64_t) nwg*((1 << ARENA_L0) + 256) + (int64_t) (cpb*span)
                                           + ((int64_t) 64 << 20);
  int64_t aln_cap   = prm->aln_cap   > 0 ? prm->aln_cap   : 4*H->nhits + 1024;
  int64_t tb_cap    = prm->trace_cap > 0 ? prm->trace_cap : (int64_t) (tpb*span) + 64*aln_cap + (1 << 20);
  { // never ask for more than one allocation can get (a free piece of the device pool, or free memory less 8 GiB)
    const int64_t lim = (int64_t) (fga_dev_largest(dev,(size_t) 8 << 30) / sizeof(int4)) - 256;
    if (lim > 0 && pool_cells > lim) pool_cells = lim;
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

  int64_t ncalls_total = 0;
  fga_unit *d_units = NULL; fga_hit *d_hits = NULL; int *d_order = NULL, *d_next = NULL, *d_wide = NULL;
  std::vector<int> wide;
  int *d_tab = NULL;
  unsigned long long *d_cnt = NULL, *d_cw = NULL;
  const int ncw = GA->nperm > 0 ? GA->nperm : 1;
  arena_lists *d_lists = NULL;
  hipError_t e;
  if ((e = fga_dmalloc(&d_lists,sizeof(arena_lists))) != hipSuccess ||
      (e = fga_dmalloc(&d_units,sizeof(fga_unit)*H->nunits)) != hipSuccess ||
      (e = fga_dmalloc(&d_hits,sizeof(fga_hit)*(H->nhits+1))) != hipSuccess ||
      (e = fga_dmalloc(&d_order,sizeof(int)*H->nunits)) != hipSuccess ||
      (e = fga_dmalloc(&d_wide,sizeof(int)*H->nunits)) != hipSuccess ||
      (e = fga_dmalloc(&d_next,sizeof(int))) != hipSuccess ||
      (e = fga_dmalloc(&d_tab,sizeof(int)*32768)) != hipSuccess ||
      (e = fga_dmalloc(&d_cw,sizeof(unsigned long long)*(size_t) ncw)) != hipSuccess ||
      (e = fga_dmalloc(&d_cnt,sizeof(unsigned long long)*32)) != hipSuccess)
    { fga_set_error("fga_extend: device allocation failed: %s",hipGetErrorString(e));
      goto fail;
    }
  hipMemcpy(d_units,H->units,sizeof(fga_unit)*H->nunits,hipMemcpyHostToDevice);
  hipMemcpy(d_hits,H->hits,sizeof(fga_hit)*H->nhits,hipMemcpyHostToDevice);
  { uint4 *r0 = NULL, *r1 = NULL, *rs = NULL;
    const unsigned gb = (unsigned) ((H->nunits + 255) / 256);
    bool ok = fga_dmalloc(&r0,sizeof(uint4)*(size_t) H->nunits) == hipSuccess && fga_dmalloc(&r1,sizeof(uint4)*(size_t) H->nunits) == hipSuccess;
    if (ok)
      { hipLaunchKernelGGL(unit_work_kernel,dim3(gb),dim3(256),0,dev->stream,(const fga_unit *) d_units,(const fga_hit *) d_hits,(int64_t) H->nunits,r0);
        ok = fga_radix_sort_u128(dev,r0,r1,H->nunits,32,WORK_BITS,&rs) == 0;
      }
    if (ok)
      { hipLaunchKernelGGL(unit_order_kernel,dim3(gb),dim3(256),0,dev->stream,(const uint4 *) rs,(int64_t) H->nunits,d_order);
        ok = hipStreamSynchronize(dev->stream) == hipSuccess && hipGetLastError() == hipSuccess;
      }
    fga_pool_free(r0); fga_pool_free(r1);
    if (!ok)
      { fga_set_error("fga_extend: ordering the units on the device failed");
        goto fail;
      }
  }
  { std::vector<int> t32(32768);
    for (int i = 0; i < 32768; i++) t32[i] = prm->table[i];
    hipMemcpy(d_tab,t32.data(),sizeof(int)*32768,hipMemcpyHostToDevice);
  }
  A.units = d_units; A.hits = d_hits; A.order = d_order; A.next = d_next; A.wide_q = d_wide;
  A.trim32 = d_tab; A.counters = d_cnt; A.ctg_waves = d_cw;

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
      A.lists = d_lists;
      hipMemsetAsync(d_lists,0,sizeof(int)*2*ARENA_NLEV,dev->stream);
      fga_dev_note_memory(dev);                  // pool + output buffers + resident inputs: a footprint peak
      hipMemsetAsync(d_next,0,sizeof(int),dev->stream);
      hipMemsetAsync(d_cnt,0,sizeof(unsigned long long)*32,dev->stream);
      hipMemsetAsync(d_cw,0,sizeof(unsigned long long)*(size_t) ncw,dev->stream);

      hipEventRecord(dev->ev0,dev->stream);
      unsigned long long hc[32];
      int64_t naln_narrow = 0;
      wide.clear();
      A.order = d_order; A.nunits = (int) H->nunits;
      if (narrow)
        { hipLaunchKernelGGL(ext_mid::extend_kernel,dim3(nwg),dim3(64),0,dev->stream,A);
          e = hipMemcpyAsync(hc,d_cnt,sizeof(hc),hipMemcpyDeviceToHost,dev->stream);
          if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
          if (e == hipSuccess) e = hipGetLastError();
          if (e != hipSuccess)
            { fga_set_error("fga_extend: kernel failed: %s",hipGetErrorString(e));
              goto fail;
            }
          naln_narrow = (int64_t) hc[0];
          const int64_t nw = (int64_t) hc[14];
          if (nw > 0 && hc[4] == 0)
            { // second launch: the full kernel over the handed-over units (longest first is no longer known: in the
              // order they were handed over); the pool starts over, the output counters go on
              wide.resize((size_t) nw);
              hipMemcpy(wide.data(),d_wide,sizeof(int)*(size_t) nw,hipMemcpyDeviceToHost);
              unsigned long long zero = 0;
              hipMemcpyAsync(d_cnt+16,&zero,sizeof(zero),hipMemcpyHostToDevice,dev->stream);
              hipMemsetAsync(d_lists,0,sizeof(int)*2*ARENA_NLEV,dev->stream);
              hipMemsetAsync(d_next,0,sizeof(int),dev->stream);
              A.order = d_wide; A.nunits = (int) nw;
              int wwg = dev->ncu * 6;
              if (wwg > nw) wwg = (int) nw;
              hipLaunchKernelGGL(ext_full::extend_kernel,dim3(wwg),dim3(64),0,dev->stream,A);
            }
        }
      else
        hipLaunchKernelGGL(ext_full::extend_kernel,dim3(nwg),dim3(64),0,dev->stream,A);
      hipEventRecord(dev->ev1,dev->stream);
      e = hipStreamSynchronize(dev->stream);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess)
        { fga_set_error("fga_extend: kernel failed: %s",hipGetErrorString(e));
          goto fail;
        }
      hipEventElapsedTime(&dev->last_ms[FGA_STAGE_EXTEND],dev->ev0,dev->ev1);
      hipMemcpy(hc,d_cnt,sizeof(hc),hipMemcpyDeviceToHost);
      R->ncalls = naln_narrow;                 // (parked: the records of handed-over units are dropped below)
      if (getenv("FGA_EXTEND_PROFILE") != NULL)
        fprintf(stderr,"extend profile: max per wavefront: steps %.2f Mcyc, unwind %.2f Mcyc, total %.2f Mcyc, waves %llu; "
                       "sum: steps %.1f Mcyc unwind %.1f Mcyc; kernel %.2f ms, %d workgroups, %lld units, %llu register->ring spills, "
                       "pool %.1f of %.1f MB; hit-box bases %.1f M, longest unit %.2f M; the longest-running wavefront: %.2f Mcyc = steps %.2f + unwind %.2f "
                       "(reverse walks %.2f) + rest, %llu calls\n",
                hc[5]*1e-6,hc[6]*1e-6,hc[7]*1e-6,hc[10],hc[8]*1e-6,hc[9]*1e-6,dev->last_ms[FGA_STAGE_EXTEND],nwg,
                (long long) H->nunits,hc[11],hc[16]*16e-6,pool_cells*16e-6,span*1e-6,span_max*1e-6,
                (double) (hc[15] >> 42)*1024e-6,(double) ((hc[15] >> 21) & 0x1fffff)*1024e-6,(double) (hc[15] & 0x1fffff)*1024e-6,
                (double) ((hc[17] >> 21) & 0x1fffff)*1024e-6,hc[17] & 0x1fffff);
#ifdef EXT_STEP_PROF
      { unsigned long long sp[8], z[8] = {0,0,0,0,0,0,0,0};
        hipMemcpyFromSymbol(sp,HIP_SYMBOL(ext_full::ext_step_prof),sizeof(sp));
        hipMemcpyToSymbol(HIP_SYMBOL(ext_full::ext_step_prof),z,sizeof(z));
        double t = 0; for (int k = 0; k < 6; k++) t += (double) sp[k];
        fprintf(stderr,"extend step sections (%% of step cycles, all wavefronts): top %.1f  neighbours+snake %.1f  trim+pebbles %.1f  best scan %.1f  clip+prune %.1f  loop %.1f   (%.0f cycles per step)\n",
                100*sp[0]/t,100*sp[1]/t,100*sp[2]/t,100*sp[3]/t,100*sp[4]/t,100*sp[5]/t,t/(double) (hc[10] ? hc[10] : 1));
      }
#endif
#ifdef EXT_MODE_PROF
      { static const char *nm[6] = { "register <= 12", "register <= 30", "register <= 60", "two blocks", "ring <= 120", "ring wider" };
        double tc = 0, tn = 0;
        for (int k = 0; k < 6; k++) { tc += (double) hc[20+k]; tn += (double) hc[26+k]; }
        for (int k = 0; k < 6; k++)
          fprintf(stderr,"extend modes: %-15s %5.1f %% of the steps, %5.1f %% of the step cycles, %7.0f cycles per step\n",nm[k],
                  100.*hc[26+k]/(tn > 0 ? tn : 1),100.*hc[20+k]/(tc > 0 ? tc : 1),hc[26+k] ? (double) hc[20+k]/(double) hc[26+k] : 0.);
      }
#endif
#ifdef EXT_WIDTH_HIST
      fprintf(stderr,"extend widths: calls by widest wave <=60 / <=120 / <=248 / wider: %llu %llu %llu %llu; wave steps in them: %llu %llu %llu %llu\n",
              hc[20],hc[21],hc[22],hc[23],hc[24],hc[25],hc[26],hc[27]);
      { const double all = (double) (hc[24] + hc[25] + hc[26] + hc[27]);
        fprintf(stderr,"extend widths, step-weighted: wave steps <=12 / <=30 / <=60 / <=120 diagonals wide: %llu %llu %llu %llu of %.0f "
                       "(%.1f %% / %.1f %% / %.1f %% / %.1f %%)\n",hc[28],hc[29],hc[30],hc[31],all,100.*hc[28]/(all > 0 ? all : 1),
                100.*hc[29]/(all > 0 ? all : 1),100.*hc[30]/(all > 0 ? all : 1),100.*hc[31]/(all > 0 ? all : 1));
      }
#endif

      R->naln = (int64_t) hc[0]; R->ntrace = (int64_t) hc[1]; ncalls_total = (int64_t) hc[2]; R->nwaves = (int64_t) hc[3];
      R->ncells = (int64_t) hc[12]; R->nbases = (int64_t) hc[13];
      // wavefronts busy on average = wave time spent in steps + unwinds, over the longest wavefront's lifetime
      R->busy_waves = hc[7] > 0 ? (double) (hc[8] + hc[9]) / (double) hc[7] : 0.;
      const bool pool_out = (hc[4] == 1), out_full = (R->naln > aln_cap || R->ntrace > tb_cap);
      if (hc[4] > 1)
        { fga_set_error("fga_extend: wave wider than the LDS ring (512 diagonals)");
          goto fail;
        }
      if (!pool_out && !out_full)
        { if (span > 0)                           // remember the real demand (beyond every wavefront's first level)
            { const double used = (double) hc[16] - (double) nwg*(1 << ARENA_L0);
              dev->ext_cells_per_base  = (used > 0. ? used : 0.) / (double) span;
              dev->ext_tbytes_per_base = (double) R->ntrace / (double) span;
            }
          break;
        }
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
  R->alns = (fga_aln *) fga_big_malloc(sizeof(fga_aln)*(R->naln+1));       // (kept between comparisons: fga_hbuf.c)
  R->tbytes = (uint8_t *) fga_big_malloc(R->ntrace+16);
  R->ctg_waves = (int64_t *) malloc(sizeof(int64_t)*(size_t) ncw);
  R->nctg_waves = GA->nperm;
  if (R->alns == NULL || R->tbytes == NULL || R->ctg_waves == NULL)
    { fga_set_error("out of memory");
      goto fail;
    }
  if (R->naln > 0)
    { const fga_aln *src = A.alns;
      uint4 *r0 = NULL, *r1 = NULL, *rs = NULL;
      fga_aln *ordered = NULL;
      // in discovery order, unless a second launch redid units (its records are told apart by their position)
      if (wide.empty() && R->naln > 1 && H->nunits < ((int64_t) 1 << 31) &&
          fga_dmalloc(&r0,sizeof(uint4)*(size_t) R->naln) == hipSuccess && fga_dmalloc(&r1,sizeof(uint4)*(size_t) R->naln) == hipSuccess &&
          fga_dmalloc(&ordered,sizeof(fga_aln)*(size_t) R->naln) == hipSuccess)
        { int ub = 1;
          while (ub < 31 && ((int64_t) 1 << ub) < H->nunits) ub += 1;
          const unsigned gb = (unsigned) ((R->naln + 255) / 256);
          hipLaunchKernelGGL(aln_key_kernel,dim3(gb),dim3(256),0,dev->stream,(const fga_aln *) A.alns,R->naln,r0);
          if (fga_radix_sort_u128(dev,r0,r1,R->naln,32,32+ub,&rs) == 0)
            { hipLaunchKernelGGL(aln_gather_kernel,dim3(gb),dim3(256),0,dev->stream,(const uint4 *) rs,(const fga_aln *) A.alns,R->naln,ordered);
              if (hipStreamSynchronize(dev->stream) == hipSuccess && hipGetLastError() == hipSuccess)
                src = ordered;
            }
        }
      hipMemcpy(R->alns,src,sizeof(fga_aln)*R->naln,hipMemcpyDeviceToHost);
      fga_pool_free(r0); fga_pool_free(r1); fga_pool_free(ordered);
    }
  if (R->ntrace > 0)
    hipMemcpy(R->tbytes,A.tbytes,R->ntrace,hipMemcpyDeviceToHost);
  hipMemcpy(R->ctg_waves,d_cw,sizeof(int64_t)*(size_t) ncw,hipMemcpyDeviceToHost);      // (a handed-over unit counts in both launches)
  if (!wide.empty())
    { // alignments the narrow launch emitted for units it later handed over: the full kernel redid those units
      const int64_t nn = R->ncalls;           // = alignments of the narrow launch (they come first)
      std::vector<char> redo((size_t) H->nunits,0);
      for (size_t q = 0; q < wide.size(); q++) redo[(size_t) wide[q]] = 1;
      int64_t o = 0;
      for (int64_t i = 0; i < R->naln; i++)
        if (!(i < nn && redo[(size_t) R->alns[i].unit]))
          R->alns[o++] = R->alns[i];
      R->naln = o;
    }
  R->ncalls = ncalls_total;
  fga_pool_free(d_units); fga_pool_free(d_hits); fga_pool_free(d_order); fga_pool_free(d_wide); fga_pool_free(d_next); fga_pool_free(d_tab); fga_pool_free(d_cnt); fga_pool_free(d_cw); fga_pool_free(d_lists);
  fga_dev_release(dev,SLOT_CELLS,A.pool);
  fga_dev_release(dev,SLOT_ALNS,A.alns); fga_dev_release(dev,SLOT_TBYTES,A.tbytes);
  *out = R;
  return 0;

fail:
  fga_pool_free(d_units); fga_pool_free(d_hits); fga_pool_free(d_order); fga_pool_free(d_wide); fga_pool_free(d_next); fga_pool_free(d_tab); fga_pool_free(d_cnt); fga_pool_free(d_cw); fga_pool_free(d_lists);
  fga_dev_release(dev,SLOT_CELLS,A.pool);
  fga_dev_release(dev,SLOT_ALNS,A.alns); fga_dev_release(dev,SLOT_TBYTES,A.tbytes);
  fga_big_free(R->alns); fga_big_free(R->tbytes); free(R->ctg_waves); free(R);
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
    arena_lists *dlists;
    std::vector<uint8_t>  pack;
    std::vector<uint16_t> trace;                  // result trace (the reference's work->points)
    std::vector<int32_t>  itrace;                 // Compute_Trace_PTS's result (the reference's work->trace): ints, one per indel
  };

struct shim_spec
  { double ave_corr; int tspace, reach; float freq[4];
    int path_ave, mscore;
  };

extern "C" void *fga_shim_New_Work_Data(void)
{ shim_work *W = new (std::nothrow) shim_work();
  if (W == NULL) { fga_set_error("out of memory"); return NULL; }
  W->dev = NULL; W->dA = W->dB = NULL; W->capA = W->capB = 0; W->pool = NULL; W->pool_cells = 0;
  W->dtrace = NULL; W->dtrace_cap = 0; W->dout = NULL; W->dcnt = NULL; W->dlists = NULL;
  const char *e = getenv("FGA_DEVICE");
  if (fga_dev_open(e != NULL ? atoi(e) : 0,&W->dev))
    { delete W; return NULL; }
  if (fga_dmalloc(&W->dout,sizeof(int)*8) != hipSuccess || fga_dmalloc(&W->dcnt,sizeof(unsigned long long)*32) != hipSuccess ||
      fga_dmalloc(&W->dlists,sizeof(arena_lists)) != hipSuccess || hipMemsetAsync(W->dlists,0,sizeof(arena_lists),W->dev->stream) != hipSuccess)
    { fga_set_error("fga_shim_New_Work_Data: device allocation failed");
      fga_pool_free(W->dout); fga_dev_close(W->dev); delete W;
      return NULL;
    }
  return W;
}

extern "C" void fga_shim_Free_Work_Data(void *work)
{ shim_work *W = (shim_work *) work;
  if (W == NULL) return;
  fga_dev_enter(W->dev);
  fga_pool_free(W->dA); fga_pool_free(W->dB); fga_pool_free(W->pool); fga_pool_free(W->dtrace); fga_pool_free(W->dout); fga_pool_free(W->dcnt); fga_pool_free(W->dlists);
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
    { fga_pool_free(*dbuf); *dbuf = NULL; *cap = 0;
      if (fga_dmalloc(dbuf,bytes + bytes/4) != hipSuccess)
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
  FGA_HIP(fga_dev_enter(W->dev));
  const int selfie = (align->aseq == align->bseq);
  if (shim_upload(W,align->aseq,align->alen,&W->dA,&W->capA)) return 1;
  if (!selfie && shim_upload(W,align->bseq,align->blen,&W->dB,&W->capB)) return 1;
  const int maxlen = align->alen > align->blen ? align->alen : align->blen;
  // pool: up to ~64 cells per 100 bases (wide waves) + the trace scratch
  const int64_t need = 64*((int64_t) maxlen/TS + 64) + (1 << ARENA_L0) + 4096;
  if (W->pool_cells < need)
    { fga_pool_free(W->pool); W->pool = NULL; W->pool_cells = 0;
      if (fga_dmalloc(&W->pool,sizeof(int4)*((size_t) need + 128)) != hipSuccess)
        { fga_set_error("fga_shim_Local_Alignment: device allocation failed");
          return 1;
        }
      W->pool_cells = need;
    }
  const int tcap = 4*(align->alen/TS + 4) + 16;
  if (W->dtrace_cap < tcap)
    { fga_pool_free(W->dtrace); W->dtrace = NULL; W->dtrace_cap = 0;
      if (fga_dmalloc(&W->dtrace,sizeof(uint16_t)*(size_t) tcap) != hipSuccess)
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
  L.G.lists = W->dlists;
  hipMemsetAsync(W->dlists,0,sizeof(int)*2*ARENA_NLEV,W->dev->stream);
  L.alen = align->alen; L.blen = selfie ? align->alen : align->blen;
  L.acomp = (align->flags & 0x2) != 0;           // ACOMP_FLAG (align.h:128)
  L.selfie = selfie;
  L.low = low; L.hgh = hgh; L.anti = anti; L.lbord = lbord; L.hbord = hbord;
  L.out = W->dout; L.trace_out = W->dtrace; L.trace_cap = W->dtrace_cap;
  hipMemsetAsync(W->dcnt,0,sizeof(unsigned long long)*32,W->dev->stream);
  hipLaunchKernelGGL(ext_full::local_alignment_one_kernel,dim3(1),dim3(64),0,W->dev->stream,L);
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


// ---------------------------------------------------------------------------------------------------
// Compute_Trace_PTS (align.h:266-267, align.c:6171-6308) and Gap_Improver (align.h:393-399, align.c:6714-7133) with the
// reference's prototypes, so that the one pair of calls every reader of a .1aln makes per record (ALNtoPAF.c:278-280,
// ALNtoPSL.c:193-197, ALNshow.c:524) can be pointed at this library with a #define.  align->path holds the trace points
// (uint16 pairs after Decompress_TraceTo16), align->aseq / bseq the NUMERIC sequences as the readers lay them out: aseq the
// whole A contig, bseq such that bseq[bbpos .. bepos) is the aligned piece of B, complemented by the caller for a
// complement alignment (ALNtoPAF.c:258-277).  On return path->trace points at the edit script (ints in the Work_Data,
// valid until the next call on it), path->tlen is its length and path->diffs its differences.  Only what the reference's
// own callers ask for is supported: trace spacing 100, mode GREEDIEST, no band (dlow > dhgh).
// One record per call: the aligned pieces are packed, uploaded and run through the batch stage fga_trace_pts (the record's
// A coordinates shifted by a multiple of the trace spacing so that only the piece travels): a parity device, not the
// fast path (fga_trace_pts[_regrouped] over a whole set).
// ---------------------------------------------------------------------------------------------------
static void shim_piece_gdb(fga_gdb *G, fga_contig *ctg, std::vector<uint8_t> &bps, const char *seq, int from, int len)
{ memset(G,0,sizeof(*G));
  memset(ctg,0,sizeof(*ctg));
  bps.assign((size_t) ((len+3) >> 2) + 16,0);
  for (int i = 0; i < len; i++)
    bps[(size_t) (i >> 2)] |= (uint8_t) ((seq[from+i] & 3) << (2*(i & 3)));
  ctg->clen = len; ctg->boff = 0; ctg->sbeg = 0; ctg->scaf = 0;
  G->ncontig = 1; G->contigs = ctg; G->maxctg = len; G->seqtot = len;
  G->bps = bps.data(); G->bpslen = (len+3) >> 2;
}

extern "C" int fga_shim_Compute_Trace_PTS(void *align_, void *work, int trace_spacing, int mode, int dlow, int dhgh)
{ shim_alignment *align = (shim_alignment *) align_;
  shim_work *W = (shim_work *) work;
  if (align == NULL || W == NULL || align->path == NULL || align->aseq == NULL || align->bseq == NULL)
    { fga_set_error("fga_shim_Compute_Trace_PTS: null argument");
      return 1;
    }
  if (trace_spacing != TS || mode != 0 || dlow <= dhgh)
    { fga_set_error("fga_shim_Compute_Trace_PTS: only trace spacing 100, mode GREEDIEST and no band (dlow > dhgh): what the "
                    "reference's readers ask for (ALNtoPAF.c:278)");
      return 1;
    }
  shim_path *P = align->path;
  if (P->abpos < 0 || P->aepos < P->abpos || P->aepos > align->alen || P->bbpos < 0 || P->bepos < P->bbpos ||
      P->bepos > align->blen || P->tlen < 0 || (P->tlen & 1) != 0 || (P->tlen > 0 && P->trace == NULL))
    { fga_set_error("fga_shim_Compute_Trace_PTS: inconsistent path");
      return 1;
    }
  FGA_HIP(fga_dev_enter(W->dev));
  const int a0 = (P->abpos / TS) * TS, b0 = P->bbpos;            // the trace points stay on multiples of 100 in A
  fga_gdb GA, GB;
  fga_contig ca, cb;
  std::vector<uint8_t> pa, pb;
  shim_piece_gdb(&GA,&ca,pa,align->aseq,a0,P->aepos - a0);
  shim_piece_gdb(&GB,&cb,pb,align->bseq,b0,P->bepos - b0);
  const int perm0 = 0;
  fga_dgenome *dga = NULL, *dgb = NULL;
  fga_traces *T = NULL;
  int rc = 1;
  std::vector<uint8_t> tb((size_t) (P->tlen > 0 ? P->tlen : 1));
  { const uint16_t *pt = (const uint16_t *) P->trace;
    for (int i = 0; i < P->tlen; i++)
      { if (pt[i] > 255)
          { fga_set_error("fga_shim_Compute_Trace_PTS: a trace point value beyond 255 (the .1aln keeps bytes)");
            return 1;
          }
        tb[(size_t) i] = (uint8_t) pt[i];
      }
  }
  fga_aln rec;
  memset(&rec,0,sizeof(rec));
  rec.tlen = P->tlen; rec.diffs = P->diffs;
  rec.abpos = P->abpos - a0; rec.aepos = P->aepos - a0; rec.bbpos = 0; rec.bepos = P->bepos - b0;
  rec.flags = 0; rec.aread = 0; rec.bread = 0; rec.toff = 0;
  fga_alns set;
  memset(&set,0,sizeof(set));
  set.naln = 1; set.ntrace = P->tlen; set.alns = &rec; set.tbytes = tb.data();
  if (fga_dgenome_upload(W->dev,&GA,&perm0,1,0,&dga) || fga_dgenome_upload(W->dev,&GB,&perm0,1,0,&dgb) ||
      fga_trace_pts(W->dev,dga,dgb,&set,TS,0,&T))
    goto done;
  { const int n = T->tlen[0];
    const int32_t *t = T->trace + T->toff[0];
    W->itrace.resize((size_t) (n > 0 ? n : 1));
    for (int i = 0; i < n; i++)                       // back to the contigs' own coordinates
      W->itrace[(size_t) i] = t[i] < 0 ? t[i] - a0 : t[i] + b0;
    P->trace = W->itrace.data();
    P->tlen = n;
    P->diffs = T->diffs[0];
  }
  rc = 0;
done:
  fga_traces_free(T);
  fga_dgenome_free(dga); fga_dgenome_free(dgb);
  return rc;
}

extern "C" int fga_shim_Gap_Improver(void *align_, void *work)
{ shim_alignment *align = (shim_alignment *) align_;
  shim_work *W = (shim_work *) work;
  if (align == NULL || W == NULL || align->path == NULL || align->aseq == NULL || align->bseq == NULL)
    { fga_set_error("fga_shim_Gap_Improver: null argument");
      return 1;
    }
  shim_path *P = align->path;
  if (P->tlen < 0 || (P->tlen > 0 && P->trace == NULL))
    { fga_set_error("fga_shim_Gap_Improver: no edit script in the path (call Compute_Trace_PTS first)");
      return 1;
    }
  // Gap_Improver looks at the whole A contig and at B between the sentinels the readers put around the aligned piece
  // (ALNtoPAF.c:258-277): the host regrouping (fga_gap_improve) reads the same through two one-contig GDBs -- A whole, B the
  // piece at its own coordinates
  fga_gdb GA, GB;
  fga_contig ca, cb;
  std::vector<uint8_t> pa, pb;
  shim_piece_gdb(&GA,&ca,pa,align->aseq,0,align->alen);
  pb.assign((size_t) ((align->blen+3) >> 2) + 16,0);
  for (int i = P->bbpos; i < P->bepos; i++)
    pb[(size_t) (i >> 2)] |= (uint8_t) ((align->bseq[i] & 3) << (2*(i & 3)));
  memset(&GB,0,sizeof(GB)); memset(&cb,0,sizeof(cb));
  cb.clen = align->blen; GB.ncontig = 1; GB.contigs = &cb; GB.maxctg = align->blen; GB.seqtot = align->blen;
  GB.bps = pb.data(); GB.bpslen = (align->blen+3) >> 2;
  fga_aln rec;
  memset(&rec,0,sizeof(rec));
  rec.diffs = P->diffs; rec.abpos = P->abpos; rec.aepos = P->aepos; rec.bbpos = P->bbpos; rec.bepos = P->bepos;
  fga_alns set;
  memset(&set,0,sizeof(set));
  set.naln = 1; set.alns = &rec;
  int64_t toff[2] = { 0, P->tlen };
  int32_t tlen = P->tlen, diffs = P->diffs;
  if ((int32_t *) P->trace != W->itrace.data())            // a script that is not ours: regrouped in our storage, like the
    { const int32_t *src = (const int32_t *) P->trace;     // reference rewrites it in the Work_Data's
      W->itrace.assign(src,src + (P->tlen > 0 ? P->tlen : 0));
      if (W->itrace.empty()) W->itrace.resize(1);
      P->trace = W->itrace.data();
    }
  fga_traces T;
  memset(&T,0,sizeof(T));
  T.naln = 1; T.ntrace = P->tlen; T.toff = toff; T.tlen = &tlen; T.diffs = &diffs; T.trace = W->itrace.data();
  if (fga_gap_improve(&GA,&GB,&set,&T))
    return 1;
  P->diffs = diffs;
  return 0;
}
