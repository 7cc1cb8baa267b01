/* fga_multi.c -- one comparison over the GPUs of one node, from ONE process: fga_multi_open / fga_multi_run / fga_multi_close
 * (and fga_run_multi = the three in a row).
 *
 * The reference keeps its parts machinery alive inside one process for the whole run: the merge threads take k-mer prefix
 * ranges (FastGA.c:2291-2321) and append every seed to the file of (own slot, Select[A contig]) (FastGA.c:5057-5134), the
 * search phase re-reads the files of one A-contig part at a time (the transpose + NPARTS loop, FastGA.c:5160-5204), and
 * la_merge puts the threads' record files together (FastGA.c:3991-4133).  Here the same cut is laid over `ndev` devices, one
 * host thread (and one HIP stream) per device that lives as long as the session, all data-path work through the stage
 * calls of fga_pipeline.c:
 *
 *   open      the GDBs (and index files, when there are any) are read ONCE and lent to every rank's session; rank r keeps
 *             only its 12-mer prefix range of both tables on device r (fga_session_open_impl, sliced), both genomes' bases
 *             whole; peer access between the devices is enabled; the rank threads then wait for work
 *   run       phase 1   rank r merges its prefix range                                         fga_session_merge
 *             exchange  seeds per A contig counted (the reference's buck[]), summed over the ranks in host memory; every rank
 *                       derives the same Select[] (fga_partition_contigs: weighted by the wave steps the contig's units took in
 *                       the session's previous run -- what phase 2's time is made of -- or by the seed counts in the first),
 *                       regroups its seeds by part on its device (fga_seeds_split_to) and rank p pulls its part's piece from
 *                       every rank's buffer: hipMemcpyPeerAsync over xGMI (fga_seeds_import_peer) -- no host staging, no
 *                       collective library
 *             phase 2   rank p sorts / chain-scans / extends its part and runs the redundancy filter on its records (all
 *                       records of a contig pair come from the part that owns the A contig)
 *                                                                           fga_session_align, fga_filter_alignments_mt
 *             finish    the surviving records are host memory of this process already: the calling thread lays the ranks'
 *                       runs out by A contig, puts ties into the reference's order from the summed per-strand seed counts and
 *                       writes the .1aln / PAF once                                          fga_session_finish_filtered
 *   close     the rank threads close their sessions and end
 * The result does not depend on ndev or on the weights (tests/test_multi_gpu.py: devices {0,0} and {0,0,0,0} -- virtual ranks
 * on one GPU --, (0,1) / (0,1,2,3) where the node has them, three runs of one session, against the reference line for
 * line).  fastga_amd/bin/FastGA reaches it with -G<n> or FGA_DEVICES; bench.py --gpus N times fga_multi_run.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"
#include "fga_session.h"

enum { CMD_IDLE = 0, CMD_RUN = 1, CMD_QUIT = 2 };

typedef struct
  { fga_multi   *M;
    int          rank;
    pthread_t    th;
    int          started;
    fga_session *Z;
    fga_run_stats st;
    fga_alns    *fil;          /* this rank's records after the redundancy filter (of the last run) */
    void        *sendbuf;      /* its seeds regrouped by part (device memory of its own device) */
    int64_t     *off;          /* [ndev+1] part p = records [off[p], off[p+1]) of sendbuf */
    int64_t     *hist;         /* [nctg] seeds per A contig of its prefix range */
    int64_t     *scount;       /* [2*nctg] the same per strand (reference tie order) */
    int64_t     *waves;        /* [nctg] wave steps per A contig of its part in the last run */
    double       open_s, phase1_s, exchange_s, phase2_s;
    int          streams;      /* this run's parts are contiguous stretches of A contigs: the rank appends its records itself */
    int64_t      nlive, cover;
    double       write_s;
  } multi_rank;

struct fga_multi
  { char        *root1, *root2;
    int          ndev;
    int         *devices;
    int          open_threads, open_flags;
    fga_mask_args masks;       /* (valid during fga_multi_open only: the caller's arrays) */
    int          have_masks;
    fga_shared_inputs shared;
    int          nctg;
    multi_rank  *R;
    fga_session *single;       /* ndev == 1: a plain session, nothing to cut */
    pthread_barrier_t bar;
    pthread_mutex_t   mu;
    pthread_cond_t    cv_cmd, cv_done;
    int          sync_made;
    int          go;           /* 0: the ranks' threads are being started, 1: all there, -1: one could not be started */
    int          cmd;
    unsigned long gen, done_gen;       /* command number the ranks have been given / have finished */
    int          failed;
    char         err[1024];
    fga_run_params Pr;         /* the ranks' parameters of the current run: their share of the host threads, no output */
    int64_t     *cost;         /* [nctg] wave steps per A contig of the previous run of this session */
    int          have_cost;
    double       load_s;
    /* the .1aln as a stream (fga_aln_stream_*): with the A contigs dealt to the ranks in original order, rank r's records all
       come before rank r+1's, and every rank formats and appends its own as soon as the ranks before it have */
    fga_run_params Pout;       /* the caller's parameters (output path, command line) */
    int          want_stream;  /* this run's output can be a stream (fga_run_can_stream) */
    fga_aln_stream *stream;    /* opened by rank 0 when the contiguous deal is balanced enough */
    int          turn;         /* the rank whose records go to the stream next */
    pthread_cond_t cv_turn;
  };

static void multi_fail(fga_multi *M, const char *where)
{ pthread_mutex_lock(&M->mu);
  if (!M->failed)
    { M->failed = 1;
      snprintf(M->err,sizeof(M->err),"%s: %s",where,fga_last_error());
    }
  pthread_mutex_unlock(&M->mu);
}

static int multi_failed(fga_multi *M)
{ int f;
  pthread_mutex_lock(&M->mu);
  f = M->failed;
  pthread_mutex_unlock(&M->mu);
  return f;
}

/* every rank passes every barrier, whatever happened: a rank that failed says so and the others skip their work */
#define STEP_BARRIER(M)  pthread_barrier_wait(&(M)->bar)

/* ---- open: this rank's slice of the tables + both genomes on its device, direct access to the other ranks' devices ---- */
static void rank_open(multi_rank *me)
{ fga_multi *M = me->M;
  const int r = me->rank, n = M->ndev;
  const double t0 = fga_wall();
  int q;
  if (fga_session_open_impl(M->root1,M->root2,M->devices[r],M->open_threads,r,n,M->open_flags,
                            M->have_masks ? &M->masks : NULL,&M->shared,&me->Z))
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
  if (!multi_failed(M))
    { const size_t nc = M->nctg > 0 ? M->nctg : 1;
      me->hist = calloc(nc,sizeof(int64_t)); me->scount = calloc(2*nc,sizeof(int64_t));
      me->waves = calloc(nc,sizeof(int64_t)); me->off = calloc(n+1,sizeof(int64_t));
      if (me->hist == NULL || me->scount == NULL || me->waves == NULL || me->off == NULL)
        { fga_set_error("out of memory"); multi_fail(M,"open"); }
    }
  STEP_BARRIER(M);
}

/* ---- one comparison: phase 1 / exchange / phase 2 + filter of this rank ---- */
static void rank_run(multi_rank *me)
{ fga_multi *M = me->M;
  const int r = me->rank, n = M->ndev, nctg = M->nctg;
  const fga_run_params *P = &M->Pr;
  fga_dseeds *seeds = NULL, *part = NULL;
  fga_alns *raw = NULL;
  int *select = NULL;
  double t0;
  int q;

  memset(&me->st,0,sizeof(me->st));
  fga_alns_free(me->fil); me->fil = NULL;
  memset(me->hist,0,sizeof(int64_t)*(nctg > 0 ? nctg : 1));
  memset(me->scount,0,sizeof(int64_t)*2*(nctg > 0 ? nctg : 1));
  memset(me->waves,0,sizeof(int64_t)*(nctg > 0 ? nctg : 1));

  /* phase 1 on the rank's prefix range; its seeds counted per A contig */
  t0 = fga_wall();
  fga_session_clear_strand_counts(me->Z);
  if (fga_session_merge(me->Z,P,0,0,&seeds,&me->st))
    multi_fail(M,"phase 1");
  else if (P->reference_threads > 0)        /* (the merge has counted them per strand for the tie order: their sums) */
    { int c;
      if (fga_session_strand_counts(me->Z,me->scount))
        multi_fail(M,"phase 1");
      else
        for (c = 0; c < nctg; c++)
          me->hist[c] = me->scount[c] + me->scount[nctg + c];
    }
  else if (fga_seeds_contig_histogram(fga_session_device(me->Z),seeds,nctg,me->hist))
    multi_fail(M,"phase 1");
  me->phase1_s = fga_wall() - t0;
  STEP_BARRIER(M);

  /* exchange: the same Select[] on every rank; seeds regrouped by part */
  t0 = fga_wall();
  if (!multi_failed(M))
    { int64_t *tot = calloc(nctg > 0 ? nctg : 1,sizeof(int64_t));
      int c;
      select = malloc(sizeof(int)*(nctg > 0 ? nctg : 1));
      if (tot == NULL || select == NULL)
        { fga_set_error("out of memory"); multi_fail(M,"exchange"); }
      else
        { const int64_t cnt = fga_seeds_count(seeds);
          if (M->have_cost)                 /* what phase 2's time is made of: the wave steps of the contig's units last time */
            for (c = 0; c < nctg; c++)
              tot[c] = M->cost[c];
          else                              /* the first run of a session: the seeds, as the reference's IDBsplit weighs bases */
            for (q = 0; q < n; q++)
              for (c = 0; c < nctg; c++)
                tot[c] += M->R[q].hist[c];
          me->streams = 0;
          if (fga_partition_contigs(tot,nctg,n,select))
            multi_fail(M,"exchange (split)");
          else if (M->want_stream)          /* the same contigs in original order, when that deal is not much worse */
            { int *inord = malloc(sizeof(int)*(nctg > 0 ? nctg : 1));
              int64_t *load = calloc(2*(size_t) n,sizeof(int64_t)), lmax = 0, omax = 0;
              if (inord != NULL && load != NULL && fga_partition_contigs_in_order(tot,me->Z->x1->perm,nctg,n,inord) == 0)
                { for (c = 0; c < nctg; c++)
                    { if ((load[select[c]] += tot[c]) > lmax) lmax = load[select[c]];
                      if ((load[n + inord[c]] += tot[c]) > omax) omax = load[n + inord[c]];
                    }
                  if (omax <= lmax + lmax/4 + 1 || fga_run_stream_forced())
                    { memcpy(select,inord,sizeof(int)*nctg);
                      me->streams = 1;
                    }
                }
              free(inord); free(load);
            }
          if (me->streams && P->reference_threads > 0)     /* the tie order wants the strand counts of ALL prefix ranges */
            { int64_t *sc = calloc(2*(size_t) (nctg > 0 ? nctg : 1),sizeof(int64_t));
              if (sc == NULL) { fga_set_error("out of memory"); multi_fail(M,"exchange"); }
              else
                { for (q = 0; q < n; q++)
                    for (c = 0; c < 2*nctg; c++)
                      sc[c] += M->R[q].scount[c];
                  if (fga_session_set_strand_counts(me->Z,sc)) multi_fail(M,"exchange");
                }
              free(sc);
            }
          if (me->streams && r == 0 && fga_session_stream_open(me->Z,&M->Pout,&M->stream))
            multi_fail(M,"output");
          if (multi_failed(M))
            ;
          else if (fga_dev_malloc(fga_session_device(me->Z),16*(size_t) (cnt > 0 ? cnt : 1),&me->sendbuf) ||
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
  if (me->sendbuf != NULL)
    { fga_dev_free(fga_session_device(me->Z),me->sendbuf); me->sendbuf = NULL; }
  me->exchange_s = fga_wall() - t0;

  /* phase 2 on the rank's part + the redundancy filter on its records */
  t0 = fga_wall();
  if (!multi_failed(M))
    { if (fga_session_align(me->Z,P,part,&raw,&me->st))
        multi_fail(M,"phase 2");
      else
        { const double tf = fga_wall();
          if (raw->ctg_waves != NULL && raw->nctg_waves == nctg)
            memcpy(me->waves,raw->ctg_waves,sizeof(int64_t)*nctg);
          if (fga_filter_alignments_mt(raw,P->nthreads,&me->fil) ||
              (me->streams && fga_session_reference_order(me->Z,P,me->fil)))
            multi_fail(M,"filter");
          me->st.filter_s += fga_wall() - tf;
        }
      part = NULL;                          /* consumed by fga_session_align */
    }
  fga_seeds_free(part);
  fga_alns_free(raw); raw = NULL;
  free(select);
  me->st.hbm_peak_bytes = fga_dev_peak_bytes(fga_session_device(me->Z));
  me->phase2_s = fga_wall() - t0;
  /* a streamed run: this rank's stretch of the .1aln, as soon as the ranks before it have written theirs (every rank takes
     its turn, whatever happened to it) */
  me->nlive = me->cover = 0; me->write_s = 0.;
  if (M->want_stream)
    { fga_aln_block *blk = NULL;
      const int mine = me->streams && M->stream != NULL && me->fil != NULL && !multi_failed(M);
      double tw = fga_wall();
      if (mine)                             /* formatted now, on this rank's threads; written when its turn has come */
        { int64_t i;
          me->nlive = me->fil->naln;
          for (i = 0; i < me->fil->naln; i++)
            me->cover += me->fil->alns[i].aepos - me->fil->alns[i].abpos;
          fga_aln_writer_threads(P->nthreads > 8 ? P->nthreads : 8);
          if (me->fil->naln > 0 && fga_aln_stream_preformats(M->stream) && fga_aln_stream_format(M->stream,me->fil,&blk))
            multi_fail(M,"output");
          me->write_s = fga_wall() - tw;
        }
      pthread_mutex_lock(&M->mu);
      while (M->turn != r)
        pthread_cond_wait(&M->cv_turn,&M->mu);
      pthread_mutex_unlock(&M->mu);
      tw = fga_wall();
      if (mine && !multi_failed(M))
        { if (blk != NULL ? fga_aln_stream_commit(M->stream,blk) : fga_aln_stream_append(M->stream,me->fil))
            multi_fail(M,"output");
          blk = NULL;
          me->write_s += fga_wall() - tw;
        }
      fga_aln_block_free(blk);
      pthread_mutex_lock(&M->mu);
      M->turn = r + 1;
      pthread_cond_broadcast(&M->cv_turn);
      pthread_mutex_unlock(&M->mu);
    }
  STEP_BARRIER(M);
}

static void *multi_rank_main(void *arg)
{ multi_rank *me = arg;
  fga_multi *M = me->M;
  unsigned long seen = 0;
  int go;

  /* all ranks or none: a session short of a rank would wait at its first barrier for ever */
  pthread_mutex_lock(&M->mu);
  while (M->go == 0)
    pthread_cond_wait(&M->cv_cmd,&M->mu);
  go = M->go;
  pthread_mutex_unlock(&M->mu);
  if (go < 0)
    return NULL;
  rank_open(me);
  for (;;)
    { int cmd;
      if (me->rank == 0)                    /* (all ranks are past the command's last barrier) */
        { pthread_mutex_lock(&M->mu);
          M->done_gen = seen;
          pthread_cond_broadcast(&M->cv_done);
          pthread_mutex_unlock(&M->mu);
        }
      pthread_mutex_lock(&M->mu);
      while (M->gen == seen)
        pthread_cond_wait(&M->cv_cmd,&M->mu);
      seen = M->gen;
      cmd = M->cmd;
      pthread_mutex_unlock(&M->mu);
      if (cmd == CMD_QUIT)
        break;
      if (multi_failed(M))                  /* (a session that did not open runs nothing; the barriers of a run are skipped by all) */
        continue;
      rank_run(me);
    }
  /* the thread that opened the session (and whose stream its buffers remember) closes it */
  if (me->sendbuf != NULL && me->Z != NULL) fga_dev_free(fga_session_device(me->Z),me->sendbuf);
  fga_session_close(me->Z); me->Z = NULL;
  return NULL;
}

/* hand the ranks a command and wait until rank 0 reports all of them past its last barrier */
static void multi_command(fga_multi *M, int cmd, int wait)
{ unsigned long g;
  pthread_mutex_lock(&M->mu);
  M->cmd = cmd;
  g = ++M->gen;
  pthread_cond_broadcast(&M->cv_cmd);
  while (wait && M->done_gen != g)
    pthread_cond_wait(&M->cv_done,&M->mu);
  pthread_mutex_unlock(&M->mu);
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
      MAXOF(merge_kernel_ms); MAXOF(sort_kernel_ms); MAXOF(extend_kernel_ms); MAXOF(chain_kernel_ms);
      MAXOF(sort_passes); MAXOF(ext_busy_waves); MAXOF(hbm_peak_bytes);
#undef MAXOF
      if (R[r].Z != NULL)
        { if (R[r].Z->load_s > S->load_s) S->load_s = R[r].Z->load_s;
          if (R[r].Z->upload_s > S->upload_s) S->upload_s = R[r].Z->upload_s;
        }
    }
}

void fga_multi_close(fga_multi *M)
{ const int dev0 = fga_dev_current_device();
  int r;
  if (M == NULL) return;
  if (M->R != NULL)
    { int any = 0;
      for (r = 0; r < M->ndev; r++) any |= M->R[r].started;
      if (any)
        multi_command(M,CMD_QUIT,0);
      for (r = 0; r < M->ndev; r++)
        if (M->R[r].started)
          pthread_join(M->R[r].th,NULL);
      for (r = 0; r < M->ndev; r++)
        { fga_alns_free(M->R[r].fil);
          free(M->R[r].off); free(M->R[r].hist); free(M->R[r].scount); free(M->R[r].waves);
        }
    }
  if (M->sync_made)
    { pthread_barrier_destroy(&M->bar);
      pthread_cond_destroy(&M->cv_cmd); pthread_cond_destroy(&M->cv_done); pthread_cond_destroy(&M->cv_turn);
      pthread_mutex_destroy(&M->mu);
    }
  fga_session_close(M->single);
  free(M->R);
  fga_gix_close(M->shared.x2); fga_gix_close(M->shared.x1);
  fga_gdb_close(M->shared.g2); fga_gdb_close(M->shared.g1);
  free(M->cost); free(M->devices); free(M->root1); free(M->root2);
  free(M);
  fga_dev_restore_device(dev0);
}

int fga_multi_open(const char *root1, const char *root2, const fga_run_params *P, int ndev, const int *devices, fga_multi **out)
{ const int dev0 = fga_dev_current_device();
  fga_multi *M;
  int r, have1, have2, self = (root2 == NULL), nhave;
  double tl;

  if (out != NULL) *out = NULL;
  if (root1 == NULL || P == NULL || devices == NULL || out == NULL || ndev < 1 || ndev > 64)
    { fga_set_error("fga_multi_open: bad argument (1 <= ndev <= 64 devices, a device list)");
      return 1;
    }
  M = calloc(1,sizeof(fga_multi));
  if (M == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  M->ndev = ndev;
  M->root1 = strdup(root1); M->root2 = self ? NULL : strdup(root2);
  M->devices = malloc(sizeof(int)*ndev);
  if (M->root1 == NULL || (!self && M->root2 == NULL) || M->devices == NULL)
    { fga_set_error("out of memory");
      goto fail;
    }
  memcpy(M->devices,devices,sizeof(int)*ndev);
  M->open_threads = P->nthreads > 0 ? P->nthreads : 8;
  M->open_flags = P->build_index ? FGA_SESSION_BUILD_INDEX : 0;
  M->masks.m1 = P->masks1; M->masks.n1 = P->nmasks1; M->masks.m2 = P->masks2; M->masks.n2 = P->nmasks2;
  M->have_masks = P->nmasks1 > 0 || P->nmasks2 > 0;

  if (ndev == 1)                          /* nothing to cut: a plain session on the one device */
    { if (fga_session_open_masked(root1,root2,devices[0],M->open_threads,M->open_flags,P->masks1,P->nmasks1,P->masks2,P->nmasks2,
                                  &M->single))
        goto fail;
      *out = M;
      fga_dev_restore_device(dev0);
      return 0;
    }

  /* the host-side inputs once for all ranks: the genomes (with their masks) and, when both exist, the index files */
  tl = fga_wall();
  have1 = !P->build_index && P->nmasks1 == 0 && fga_gix_files_exist(root1);
  have2 = self ? have1 : (!P->build_index && P->nmasks2 == 0 && fga_gix_files_exist(root2));
  if (fga_gdb_open(root1,&M->shared.g1) || (P->nmasks1 > 0 && fga_gdb_apply_masks(M->shared.g1,P->masks1,P->nmasks1)))
    goto fail;
  if (!self && (fga_gdb_open(root2,&M->shared.g2) || (P->nmasks2 > 0 && fga_gdb_apply_masks(M->shared.g2,P->masks2,P->nmasks2))))
    goto fail;
  if (have1 && have2)
    { if (fga_gix_open(root1,&M->shared.x1) || (!self && fga_gix_open(root2,&M->shared.x2)))
        goto fail;
    }
  else if (have1 != have2)                /* one genome has index files, the other (masked, or without files) has not: a
                                             sliced session cuts its ranges from file counts OR device counts -- both built */
    M->open_flags |= FGA_SESSION_BUILD_INDEX;
  M->load_s = fga_wall() - tl;
  nhave = fga_dev_device_count();         /* (after the inputs, like fga_run: a missing genome or mask is reported first) */
  if (nhave <= 0)
    { fga_set_error("no HIP device available: libfastga_amd has no CPU fallback");
      goto fail;
    }
  for (r = 0; r < ndev; r++)
    if (devices[r] < 0 || devices[r] >= nhave)
      { fga_set_error("fga_multi_open: device %d out of range (have %d)",devices[r],nhave);
        goto fail;
      }
  M->R = calloc(ndev,sizeof(multi_rank));
  if (M->R == NULL)
    { fga_set_error("out of memory");
      goto fail;
    }
  pthread_mutex_init(&M->mu,NULL);
  pthread_cond_init(&M->cv_cmd,NULL); pthread_cond_init(&M->cv_done,NULL); pthread_cond_init(&M->cv_turn,NULL);
  pthread_barrier_init(&M->bar,NULL,ndev);
  M->sync_made = 1;
  M->done_gen = 1;                        /* (the open counts as command 0: done_gen is set to 0 when all ranks are through) */
  for (r = 0; r < ndev; r++)
    { M->R[r].M = M; M->R[r].rank = r; }
  for (r = 0; r < ndev; r++)
    { if (pthread_create(&M->R[r].th,NULL,multi_rank_main,&M->R[r]) != 0)
        break;
      M->R[r].started = 1;
    }
  pthread_mutex_lock(&M->mu);
  M->go = (r == ndev) ? 1 : -1;
  pthread_cond_broadcast(&M->cv_cmd);
  pthread_mutex_unlock(&M->mu);
  if (r < ndev)
    { int q;
      for (q = 0; q < r; q++) { pthread_join(M->R[q].th,NULL); M->R[q].started = 0; }
      fga_set_error("fga_multi_open: cannot start a thread for rank %d",r);
      goto fail;
    }
  pthread_mutex_lock(&M->mu);             /* wait for the open to be through on every rank */
  while (M->done_gen != 0)
    pthread_cond_wait(&M->cv_done,&M->mu);
  pthread_mutex_unlock(&M->mu);
  M->masks.m1 = M->masks.m2 = NULL; M->masks.n1 = M->masks.n2 = 0;       /* (the caller's arrays: not kept) */
  if (M->failed)
    { char e[1024];
      snprintf(e,sizeof(e),"%s",M->err);
      fga_multi_close(M);
      fga_set_error("fga_multi_open (%d devices): %s",ndev,e);
      fga_dev_restore_device(dev0);
      return 1;
    }
  M->cost = calloc(M->nctg > 0 ? M->nctg : 1,sizeof(int64_t));
  if (M->cost == NULL)
    { fga_multi_close(M);
      fga_set_error("out of memory");
      fga_dev_restore_device(dev0);
      return 1;
    }
  *out = M;
  fga_dev_restore_device(dev0);
  return 0;

fail:
  { char e[1024];
    snprintf(e,sizeof(e),"%s",fga_last_error());
    fga_multi_close(M);
    fga_set_error("%s",e);
  }
  fga_dev_restore_device(dev0);
  return 1;
}

int fga_multi_ndev(const fga_multi *M) { return M == NULL ? 0 : M->ndev; }

/* what bench.py prices the merge launches with: entry bytes of both tables over all ranks' slices (N1 E1 + N2 E2), the
   reference's seed record width, the bases of the two genomes */
int fga_multi_info(const fga_multi *M, int64_t *table_bytes, int *seed_bytes, int64_t *bases1, int64_t *bases2)
{ int r;
  int64_t tb = 0;
  if (M == NULL)
    { fga_set_error("fga_multi_info: null argument");
      return 1;
    }
  if (M->single != NULL)
    { if (table_bytes != NULL) *table_bytes = fga_session_table_bytes(M->single);
      if (seed_bytes != NULL) *seed_bytes = fga_session_seed_bytes(M->single);
      if (bases1 != NULL) *bases1 = fga_session_bases(M->single,0);
      if (bases2 != NULL) *bases2 = fga_session_bases(M->single,1);
      return 0;
    }
  for (r = 0; r < M->ndev; r++)
    tb += fga_session_table_bytes(M->R[r].Z);
  if (table_bytes != NULL) *table_bytes = tb;
  if (seed_bytes != NULL) *seed_bytes = fga_session_seed_bytes(M->R[0].Z);
  if (bases1 != NULL) *bases1 = fga_session_bases(M->R[0].Z,0);
  if (bases2 != NULL) *bases2 = fga_session_bases(M->R[0].Z,1);
  return 0;
}

int fga_multi_run(fga_multi *M, const fga_run_params *P, fga_run_stats *S)
{ const int dev0 = fga_dev_current_device();
  fga_run_stats st;
  int r, c, rc = 1, ndev;
  const double t0 = fga_wall();

  memset(&st,0,sizeof(st));
  if (S != NULL) *S = st;
  if (M == NULL || P == NULL)
    { fga_set_error("fga_multi_run: null argument");
      return 1;
    }
  ndev = M->ndev;
  if (M->single != NULL)
    { fga_run_params Q = *P;
      if (M->have_masks) Q.soft_mask = 1;   /* masks named: the comparison runs with soft masking on (FastGA.c:4580) */
      rc = fga_session_run(M->single,&Q,S);
      fga_dev_restore_device(dev0);
      return rc;
    }

  M->Pr = *P;
  M->Pr.out_path = NULL; M->Pr.paf_path = NULL;
  /* a rank's host threads do its chain tails, its redundancy filter and its record formatting: at least four each */
  M->Pr.nthreads = (P->nthreads > 0 ? P->nthreads : 8) / ndev;
  if (M->Pr.nthreads < 4) M->Pr.nthreads = 4;
  if (M->have_masks) M->Pr.soft_mask = 1;
  M->Pout = *P;
  M->want_stream = fga_run_can_stream(P);
  M->stream = NULL; M->turn = 0;
  multi_command(M,CMD_RUN,1);
  if (M->failed)
    { fga_set_error("fga_multi_run (%d devices): %s",ndev,M->err);
      goto done;
    }

  /* the cost of every A contig's units in this run: the weights of the next run's partition */
  for (c = 0; c < M->nctg; c++)
    { int64_t w = 0;
      for (r = 0; r < ndev; r++) w += M->R[r].waves[c];
      M->cost[c] = w;
    }
  M->have_cost = 1;

  /* ---- a streamed run: every rank has appended its stretch; the footer ---- */
  if (M->stream != NULL)
    { const double tw = fga_wall();
      st.load_s = M->load_s;
      sum_stats(&st,M->R,ndev);
      for (r = 0; r < ndev; r++)
        { st.nlive += M->R[r].nlive; st.cover += M->R[r].cover;
          if (M->R[r].write_s > st.write_s) st.write_s = M->R[r].write_s;
        }
      if (fga_aln_stream_close(M->stream,1))
        { M->stream = NULL;
          goto done;
        }
      M->stream = NULL;
      st.write_s += fga_wall() - tw;
      st.streamed_parts = ndev;
    }
  else
  /* ---- finish on rank 0's session: the ranks' filtered runs by A contig, the reference's tie order, one .1aln ---- */
  { const fga_alns **sets = calloc(ndev,sizeof(fga_alns *));
    fga_run_params Pf = *P;
    int bad = (sets == NULL);
    if (bad) fga_set_error("out of memory");
    if (M->have_masks) Pf.soft_mask = 1;
    if (!bad && P->reference_threads > 0)
      { const int nc = M->nctg;
        int64_t *sc = calloc(2*(size_t) (nc > 0 ? nc : 1),sizeof(int64_t));
        if (sc == NULL) { fga_set_error("out of memory"); bad = 1; }
        else
          { for (r = 0; r < ndev; r++)
              for (c = 0; c < 2*nc; c++)
                sc[c] += M->R[r].scount[c];
            bad = fga_session_set_strand_counts(M->R[0].Z,sc);
          }
        free(sc);
      }
    for (r = 0; r < ndev && !bad; r++)
      sets[r] = M->R[r].fil;
    st.load_s = M->load_s;
    sum_stats(&st,M->R,ndev);
    if (!bad && fga_session_finish_filtered(M->R[0].Z,&Pf,sets,ndev,&st)) bad = 1;
    free(sets);
    if (bad) goto done;
  }
  st.nparts = ndev;
  st.bases1 = M->shared.g1->seqtot; st.bases2 = M->shared.g2 == NULL ? M->shared.g1->seqtot : M->shared.g2->seqtot;
  st.phase23_s = fga_wall() - t0 - st.trace_s - st.paf_s;
  if (getenv("FGA_TIMING") != NULL && atoi(getenv("FGA_TIMING")) != 0)
    for (r = 0; r < ndev; r++)
      fprintf(stderr,"[fga timing] rank %d on device %d: open %.3f  phase 1 %.3f  exchange %.3f  phase 2 %.3f s (%lld wave steps)\n",
              r,M->devices[r],M->R[r].open_s,M->R[r].phase1_s,M->R[r].exchange_s,M->R[r].phase2_s,(long long) M->R[r].st.nwaves);
  rc = 0;

done:
  if (M->stream != NULL)                  /* (a failed run leaves no file) */
    { fga_aln_stream_close(M->stream,0); M->stream = NULL; }
  for (r = 0; r < ndev; r++)              /* the records have been written: only the session stays */
    { fga_alns_free(M->R[r].fil); M->R[r].fil = NULL; }
  if (S != NULL) *S = st;
  fga_dev_restore_device(dev0);
  return rc;
}

/* per-rank figures of the last fga_multi_run (bench.py's per_rank block): seconds of phase 1 / exchange / phase 2 (incl. the
   filter), the extension kernel's ms and wave steps, the seeds of the rank's part */
int fga_multi_rank_stats(const fga_multi *M, int rank, double *seconds3, double *extend_kernel_ms, int64_t *wave_steps)
{ if (M == NULL || M->R == NULL || rank < 0 || rank >= M->ndev)
    { fga_set_error("fga_multi_rank_stats: no such rank");
      return 1;
    }
  if (seconds3 != NULL)
    { seconds3[0] = M->R[rank].phase1_s; seconds3[1] = M->R[rank].exchange_s; seconds3[2] = M->R[rank].phase2_s; }
  if (extend_kernel_ms != NULL) *extend_kernel_ms = M->R[rank].st.extend_kernel_ms;
  if (wave_steps != NULL) *wave_steps = M->R[rank].st.nwaves;
  return 0;
}

int fga_run_multi(const char *root1, const char *root2, const fga_run_params *P, int ndev, const int *devices,
                  fga_run_stats *S)
{ fga_multi *M = NULL;
  fga_run_stats st;
  int rc;
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
  if (fga_multi_open(root1,root2,P,ndev,devices,&M))
    return 1;
  rc = fga_multi_run(M,P,S);
  if (rc != 0)
    { char e[1024];
      snprintf(e,sizeof(e),"%s",fga_last_error());
      fga_multi_close(M);
      fga_set_error("%s",e);
      return rc;
    }
  fga_multi_close(M);
  return 0;
}
