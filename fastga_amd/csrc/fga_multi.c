/* fga_multi.c -- one comparison over the GPUs of one node, from ONE process: fga_run_multi.
 *
 * The reference runs its whole parts machinery inside one process: the merge threads take k-mer prefix ranges
 * (FastGA.c:2291-2321) and append every seed to the file of (own slot, Select[A contig]) (FastGA.c:5057-5134), the search
 * phase re-reads the files of one A-contig part at a time (the transpose + NPARTS loop, FastGA.c:5160-5204), and la_merge
 * puts the threads' record files together (FastGA.c:3991-4133).  Here the same cut is laid over `ndev` devices, one host
 * thread (and one HIP stream) per device, all data-path work through the stage calls of fga_pipeline.c:
 *
 *   open      the GDBs (and index files, when there are any) are read ONCE and lent to every rank's session; rank r keeps
 *             only its 12-mer prefix range of both tables on device r (fga_session_open_impl, sliced), both genomes' bases whole
 *   phase 1   rank r merges its prefix range                                         fga_session_merge
 *   exchange  seeds per A contig counted (the reference's buck[]), summed over the ranks in host memory; every rank derives
 *             the same Select[] from the sums (fga_partition_contigs), regroups its seeds by part on its device
 *             (fga_seeds_split_to) and rank p pulls its part's piece from every rank's buffer: hipMemcpyPeerAsync over xGMI
 *             (fga_seeds_import_peer) -- no host staging, no collective library
 *   phase 2   rank p sorts / chain-scans / extends its part and runs the redundancy filter on its records (all records of a
 *             contig pair come from the part that owns the A contig)         fga_session_align, fga_filter_alignments_mt
 *   finish    the surviving records are host memory of this process already: rank 0 lays the ranks' runs out by A contig,
 *             puts ties into the reference's order from the summed per-strand seed counts and writes the .1aln / PAF once
 *                                                                                      fga_session_finish_filtered
 * The result does not depend on ndev (tests/test_multi_gpu.py: devices {0,0} and {0,0,0,0} -- virtual ranks on one GPU --
 * against the reference line for line).  fastga_amd/bin/FastGA reaches it with -G<n> or FGA_DEVICES.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"
#include "fga_session.h"

typedef struct multi_run multi_run;

typedef struct
  { multi_run   *M;
    int          rank;
    pthread_t    th;
    int          started;
    fga_session *Z;
    fga_run_stats st;
    fga_alns    *fil;          /* this rank's records after the redundancy filter */
    void        *sendbuf;      /* its seeds regrouped by part (device memory of its own device) */
    int64_t     *off;          /* [ndev+1] part p = records [off[p], off[p+1]) of sendbuf */
    int64_t     *hist;         /* [nctg] seeds per A contig of its prefix range */
    int64_t     *scount;       /* [2*nctg] the same per strand (reference tie order) */
    double       open_s, phase1_s, exchange_s, phase2_s;
  } multi_rank;

struct multi_run
  { const char *root1, *root2;
    const fga_run_params *P;
    fga_run_params Pr;         /* the ranks' copy: their share of the host threads, no output */
    int          ndev;
    const int   *devices;
    fga_shared_inputs shared;
    fga_mask_args masks;
    int          have_masks;
    int          nctg;
    multi_rank  *R;
    pthread_barrier_t bar;
    pthread_mutex_t   mu;
    pthread_cond_t    cv;
    int          go;           /* 0: the ranks' threads are being started, 1: all there, run, -1: one could not be started */
    int          failed;
    char         err[1024];
  };

static void multi_fail(multi_run *M, const char *where)
{ pthread_mutex_lock(&M->mu);
  if (!M->failed)
    { M->failed = 1;
      snprintf(M->err,sizeof(M->err),"%s: %s",where,fga_last_error());
    }
  pthread_mutex_unlock(&M->mu);
}

static int multi_failed(multi_run *M)
{ int f;
  pthread_mutex_lock(&M->mu);
  f = M->failed;
  pthread_mutex_unlock(&M->mu);
  return f;
}

/* every rank passes every barrier, whatever happened: a rank that failed says so and the others skip their work */
#define STEP_BARRIER(M)  pthread_barrier_wait(&(M)->bar)

static void *multi_rank_main(void *arg)
{ multi_rank *me = arg;
  multi_run *M = me->M;
  const int r = me->rank, n = M->ndev;
  const fga_run_params *P = &M->Pr;
  fga_dseeds *seeds = NULL, *part = NULL;
  fga_alns *raw = NULL;
  int *select = NULL;
  double t0;
  int q;

  /* all ranks or none: a run short of a rank would wait at its first barrier for ever */
  pthread_mutex_lock(&M->mu);
  while (M->go == 0)
    pthread_cond_wait(&M->cv,&M->mu);
  q = M->go;
  pthread_mutex_unlock(&M->mu);
  if (q < 0)
    return NULL;

  /* ---- open: this rank's slice of the tables + both genomes on its device ---- */
  t0 = fga_wall();
  if (fga_session_open_impl(M->root1,M->root2,M->devices[r],M->P->nthreads > 0 ? M->P->nthreads : 8,r,n,
                            M->P->build_index ? FGA_SESSION_BUILD_INDEX : 0,M->have_masks ? &M->masks : NULL,&M->shared,&me->Z))
    multi_fail(M,"open");
  else
    for (q = 0; q < n; q++)
      if (fga_dev_enable_peer(fga_session_device(me->Z),M->devices[q]))
        { multi_fail(M,"peer access"); break; }
  me->open_s = fga_wall() - t0;
  STEP_BARRIER(M);
  if (!multi_failed(M) && r == 0)
    M->nctg = fga_session_nctg(me->Z);
  STEP_BARRIER(M);

  /* ---- phase 1 on the rank's prefix range; its seeds counted per A contig ---- */
  t0 = fga_wall();
  if (!multi_failed(M))
    { const int nctg = M->nctg;
      me->hist = calloc(nctg > 0 ? nctg : 1,sizeof(int64_t));
      me->scount = calloc(2*(size_t) (nctg > 0 ? nctg : 1),sizeof(int64_t));
      me->off = calloc(n+1,sizeof(int64_t));
      if (me->hist == NULL || me->scount == NULL || me->off == NULL)
        { fga_set_error("out of memory"); multi_fail(M,"phase 1"); }
      else
        { fga_session_clear_strand_counts(me->Z);
          if (fga_session_merge(me->Z,P,0,0,&seeds,&me->st) ||
              fga_seeds_contig_histogram(fga_session_device(me->Z),seeds,nctg,me->hist) ||
              (P->reference_threads > 0 && fga_session_strand_counts(me->Z,me->scount)))
            multi_fail(M,"phase 1");
        }
    }
  me->phase1_s = fga_wall() - t0;
  STEP_BARRIER(M);

  /* ---- exchange: the same Select[] on every rank from the summed counts; seeds regrouped by part ---- */
  t0 = fga_wall();
  if (!multi_failed(M))
    { const int nctg = M->nctg;
      int64_t *tot = calloc(nctg > 0 ? nctg : 1,sizeof(int64_t));
      int c;
      select = malloc(sizeof(int)*(nctg > 0 ? nctg : 1));
      if (tot == NULL || select == NULL)
        { fga_set_error("out of memory"); multi_fail(M,"exchange"); }
      else
        { const int64_t cnt = fga_seeds_count(seeds);
          for (q = 0; q < n; q++)
            for (c = 0; c < nctg; c++)
              tot[c] += M->R[q].hist[c];
          if (fga_partition_contigs(tot,nctg,n,select) ||
              fga_dev_malloc(fga_session_device(me->Z),16*(size_t) (cnt > 0 ? cnt : 1),&me->sendbuf) ||
              fga_seeds_split_to(fga_session_device(me->Z),seeds,select,nctg,n,me->sendbuf,me->off))
            multi_fail(M,"exchange (split)");
        }
      free(tot);
    }
  fga_seeds_free(seeds); seeds = NULL;
  STEP_BARRIER(M);
  if (!multi_failed(M))
    { const void **src = malloc(sizeof(void *)*n);
      int64_t *cnt = malloc(sizeof(int64_t)*n);
      int *ids = malloc(sizeof(int)*n);
      if (src == NULL || cnt == NULL || ids == NULL)
        { fga_set_error("out of memory"); multi_fail(M,"exchange"); }
      else
        { for (q = 0; q < n; q++)           /* rank q's piece for part r, starting with the rank's own */
            { const int s = (r + q) % n;
              src[q] = (const char *) M->R[s].sendbuf + 16*(size_t) M->R[s].off[r];
              cnt[q] = M->R[s].off[r+1] - M->R[s].off[r];
              ids[q] = M->devices[s];
            }
          if (fga_seeds_import_peer(fga_session_device(me->Z),src,ids,cnt,n,&part))
            multi_fail(M,"exchange (import)");
        }
      free(src); free(cnt); free(ids);
    }
  STEP_BARRIER(M);                          /* every part has been pulled: the send buffers can go */
  if (me->sendbuf != NULL && me->Z != NULL)
    { fga_dev_free(fga_session_device(me->Z),me->sendbuf); me->sendbuf = NULL; }
  me->exchange_s = fga_wall() - t0;

  /* ---- phase 2 on the rank's part + the redundancy filter on its records ---- */
  t0 = fga_wall();
  if (!multi_failed(M))
    { if (fga_session_align(me->Z,P,part,&raw,&me->st))
        multi_fail(M,"phase 2");
      else
        { const double tf = fga_wall();
          if (fga_filter_alignments_mt(raw,P->nthreads,&me->fil))
            multi_fail(M,"filter");
          me->st.filter_s += fga_wall() - tf;
        }
      part = NULL;                          /* consumed by fga_session_align */
    }
  fga_seeds_free(part);
  fga_alns_free(raw);
  free(select);
  if (me->Z != NULL)
    me->st.hbm_peak_bytes = fga_dev_peak_bytes(fga_session_device(me->Z));
  me->phase2_s = fga_wall() - t0;
  STEP_BARRIER(M);
  return NULL;
}

static void sum_stats(fga_run_stats *S, const multi_rank *R, int n)
{ int r;
  for (r = 0; r < n; r++)
    { const fga_run_stats *s = &R[r].st;
      S->nseeds += s->nseeds; S->seed_len_sum += s->seed_len_sum; S->nhits += s->nhits; S->nunits += s->nunits;
      S->nalns += s->nalns; S->ncalls += s->ncalls; S->nwaves += s->nwaves;
      S->ext_cells += s->ext_cells; S->ext_bases += s->ext_bases; S->ext_trace += s->ext_trace;
      S->sort_keys += s->sort_keys;
      /* the ranks run side by side: a stage takes as long as its slowest rank */
#define MAXOF(f) if (s->f > S->f) S->f = s->f
      MAXOF(merge_s); MAXOF(sort_s); MAXOF(chain_s); MAXOF(extend_s); MAXOF(filter_s);
      MAXOF(merge_kernel_ms); MAXOF(sort_kernel_ms); MAXOF(extend_kernel_ms);
      MAXOF(sort_passes); MAXOF(ext_busy_waves); MAXOF(hbm_peak_bytes);
#undef MAXOF
      if (R[r].Z != NULL)
        { if (R[r].Z->load_s > S->load_s) S->load_s = R[r].Z->load_s;
          if (R[r].Z->upload_s > S->upload_s) S->upload_s = R[r].Z->upload_s;
        }
    }
}

int fga_run_multi(const char *root1, const char *root2, const fga_run_params *P, int ndev, const int *devices,
                  fga_run_stats *S)
{ multi_run M;
  fga_run_stats st;
  int r, rc = 1, have1, have2, self = (root2 == NULL), nhave;
  double t0 = fga_wall(), tl;

  memset(&st,0,sizeof(st));
  if (S != NULL) *S = st;
  if (root1 == NULL || P == NULL || devices == NULL || ndev < 1 || ndev > 64)
    { fga_set_error("fga_run_multi: bad argument (1 <= ndev <= 64 devices, a device list)");
      return 1;
    }
  if (ndev == 1)                          /* nothing to cut */
    { fga_run_params Q = *P;
      Q.device = devices[0];
      return fga_run(root1,root2,&Q,S);
    }

  memset(&M,0,sizeof(M));
  M.root1 = root1; M.root2 = root2; M.P = P; M.ndev = ndev; M.devices = devices;
  M.Pr = *P;
  M.Pr.out_path = NULL; M.Pr.paf_path = NULL;
  M.Pr.nthreads = (P->nthreads > 0 ? P->nthreads : 8) / ndev;
  if (M.Pr.nthreads < 1) M.Pr.nthreads = 1;
  M.masks.m1 = P->masks1; M.masks.n1 = P->nmasks1; M.masks.m2 = P->masks2; M.masks.n2 = P->nmasks2;
  M.have_masks = P->nmasks1 > 0 || P->nmasks2 > 0;
  if (M.have_masks)                       /* masks named: the comparison runs with soft masking on (FastGA.c:4580) */
    M.Pr.soft_mask = 1;
  M.R = calloc(ndev,sizeof(multi_rank));
  if (M.R == NULL)
    { fga_set_error("out of memory");
      return 1;
    }

  /* the host-side inputs once for all ranks: the genomes (with their masks) and, when both exist, the index files */
  tl = fga_wall();
  have1 = !P->build_index && P->nmasks1 == 0 && fga_gix_files_exist(root1);
  have2 = self ? have1 : (!P->build_index && P->nmasks2 == 0 && fga_gix_files_exist(root2));
  if (fga_gdb_open(root1,&M.shared.g1) || (P->nmasks1 > 0 && fga_gdb_apply_masks(M.shared.g1,P->masks1,P->nmasks1)))
    goto done;
  if (!self && (fga_gdb_open(root2,&M.shared.g2) || (P->nmasks2 > 0 && fga_gdb_apply_masks(M.shared.g2,P->masks2,P->nmasks2))))
    goto done;
  if (have1 && have2)
    { if (fga_gix_open(root1,&M.shared.x1) || (!self && fga_gix_open(root2,&M.shared.x2)))
        goto done;
    }
  st.load_s = fga_wall() - tl;
  nhave = fga_dev_device_count();         /* (after the inputs, like fga_run: a missing genome or mask is reported first) */
  if (nhave <= 0)
    { fga_set_error("no HIP device available: libfastga_amd has no CPU fallback");
      goto done;
    }
  for (r = 0; r < ndev; r++)
    if (devices[r] < 0 || devices[r] >= nhave)
      { fga_set_error("fga_run_multi: device %d out of range (have %d)",devices[r],nhave);
        goto done;
      }

  pthread_mutex_init(&M.mu,NULL);
  pthread_cond_init(&M.cv,NULL);
  pthread_barrier_init(&M.bar,NULL,ndev);
  for (r = 0; r < ndev; r++)
    { M.R[r].M = &M; M.R[r].rank = r; }
  for (r = 1; r < ndev; r++)
    { if (pthread_create(&M.R[r].th,NULL,multi_rank_main,&M.R[r]) != 0)
        break;
      M.R[r].started = 1;
    }
  pthread_mutex_lock(&M.mu);
  M.go = (r == ndev) ? 1 : -1;
  pthread_cond_broadcast(&M.cv);
  pthread_mutex_unlock(&M.mu);
  if (r == ndev)
    multi_rank_main(&M.R[0]);             /* rank 0 on the calling thread */
  else
    { M.failed = 1;
      snprintf(M.err,sizeof(M.err),"cannot start a thread for rank %d",r);
    }
  for (r = 1; r < ndev; r++)
    if (M.R[r].started)
      pthread_join(M.R[r].th,NULL);
  pthread_barrier_destroy(&M.bar);
  pthread_cond_destroy(&M.cv);
  pthread_mutex_destroy(&M.mu);
  if (M.failed)
    { fga_set_error("fga_run_multi (%d devices): %s",ndev,M.err);
      goto done;
    }

  /* ---- finish on rank 0's session: the ranks' filtered runs by A contig, the reference's tie order, one .1aln ---- */
  { const fga_alns **sets = calloc(ndev,sizeof(fga_alns *));
    fga_run_params Pf = *P;
    int bad = (sets == NULL);
    if (bad) fga_set_error("out of memory");
    if (M.have_masks) Pf.soft_mask = 1;
    if (!bad && P->reference_threads > 0)
      { const int nc = M.nctg;
        int64_t *sc = calloc(2*(size_t) (nc > 0 ? nc : 1),sizeof(int64_t));
        int c;
        if (sc == NULL) { fga_set_error("out of memory"); bad = 1; }
        else
          { for (r = 0; r < ndev; r++)
              for (c = 0; c < 2*nc; c++)
                sc[c] += M.R[r].scount[c];
            bad = fga_session_set_strand_counts(M.R[0].Z,sc);
          }
        free(sc);
      }
    for (r = 0; r < ndev && !bad; r++)
      sets[r] = M.R[r].fil;
    sum_stats(&st,M.R,ndev);
    if (!bad && fga_session_finish_filtered(M.R[0].Z,&Pf,sets,ndev,&st)) bad = 1;
    free(sets);
    if (bad) goto done;
  }
  st.nparts = ndev;
  st.bases1 = M.shared.g1->seqtot; st.bases2 = self ? M.shared.g1->seqtot : M.shared.g2->seqtot;
  st.phase23_s = fga_wall() - t0 - st.trace_s - st.paf_s;
  if (getenv("FGA_TIMING") != NULL && atoi(getenv("FGA_TIMING")) != 0)
    for (r = 0; r < ndev; r++)
      fprintf(stderr,"[fga timing] rank %d on device %d: open %.3f  phase 1 %.3f  exchange %.3f  phase 2 %.3f s\n",
              r,devices[r],M.R[r].open_s,M.R[r].phase1_s,M.R[r].exchange_s,M.R[r].phase2_s);
  rc = 0;

done:
  if (M.R != NULL)
    for (r = 0; r < ndev; r++)
      { if (M.R[r].sendbuf != NULL && M.R[r].Z != NULL) fga_dev_free(fga_session_device(M.R[r].Z),M.R[r].sendbuf);
        fga_alns_free(M.R[r].fil);
        fga_session_close(M.R[r].Z);
        free(M.R[r].off); free(M.R[r].hist); free(M.R[r].scount);
      }
  free(M.R);
  fga_gix_close(M.shared.x2); fga_gix_close(M.shared.x1);
  fga_gdb_close(M.shared.g2); fga_gdb_close(M.shared.g1);
  if (S != NULL) *S = st;
  return rc;
}
