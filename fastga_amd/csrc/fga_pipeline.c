/* fga_pipeline.c -- the FastGA hot path end to end on one GPU: the span of the reference's "Total Resources"
 * line (FastGA.c:4828-4829, 5263-5264): prebuilt GDB + GIX in, .1aln out.
 *
 *   phase 1  adaptive seed merge          adaptamer_merge / self_adaptamer_merge (FastGA.c:2281, 2496)   GPU
 *   phase 2  seed -> diagonal record sort reimport_thread + rmsd_sort (FastGA.c:2641, RSDsort.c:292)     GPU
 *            chain detection              align_contigs scan (FastGA.c:3016-3176)                        host threads
 *            wave extension               Local_Alignment loop (FastGA.c:3227-3341, align.c:1423)        GPU
 *            redundancy filter            FastGA.c:3405-3694                                             host
 *   phase 3  order + emit                 la_sort / la_merge (FastGA.c:3800-4133)                        host
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"
#include "fga_session.h"

/* the redundancy filter of one part's records on a thread of its own, while the next part's kernels run -- and, when the parts
   are contiguous in the output's order (A contigs in original order), the part's stretch of the .1aln behind it: the
   reference's tie order, the records formatted and appended to the stream as soon as the parts before have been */
typedef struct
  { pthread_mutex_t mu;
    pthread_cond_t  cv;
    int             turn;        /* the part whose records go to the stream next */
    int             failed;
  } part_chain;

static int reference_order(fga_session *Z, const fga_run_params *P, fga_alns *fin);

/* can the .1aln of this run be written as a stream of A-contig stretches (fga_aln_stream_*)?  Not with PAF / PSL output
   (which wants the whole set), not the text form; FGA_STREAM_PARTS=0 switches it off */
int fga_run_can_stream(const fga_run_params *P)
{ return P->out_path != NULL && P->paf_path == NULL &&
         !(getenv("FGA_ALN_ASCII") != NULL && atoi(getenv("FGA_ALN_ASCII")) != 0) &&
         !(getenv("FGA_STREAM_PARTS") != NULL && atoi(getenv("FGA_STREAM_PARTS")) == 0);
}

/* FGA_STREAM_PARTS=2: stream whatever the balance of the contiguous deal (tests) */
int fga_run_stream_forced(void)
{ return getenv("FGA_STREAM_PARTS") != NULL && atoi(getenv("FGA_STREAM_PARTS")) >= 2; }

int fga_session_stream_open(fga_session *Z, const fga_run_params *P, fga_aln_stream **out)
{ char *n1 = NULL, *n2 = NULL;
  int rc;
  if (asprintf(&n1,"%s",Z->g1->path) < 0) n1 = NULL;
  if (!Z->self && asprintf(&n2,"%s",Z->g2->path) < 0) n2 = NULL;
  rc = fga_aln_stream_open(P->out_path,Z->g1,Z->self ? NULL : Z->g2,100,n1 ? n1 : "genome1",n2,
                           P->command_line ? P->command_line : "FastGA",out);
  free(n1); free(n2);
  return rc;
}

int fga_session_reference_order(fga_session *Z, const fga_run_params *P, fga_alns *set)
{ return P->reference_threads > 0 ? reference_order(Z,P,set) : 0; }

typedef struct
  { const fga_alns *in;
    fga_alns *out;
    int nthreads, rc, started, part;
    double seconds, write_s;
    char *err;
    pthread_t th;
    /* streaming (stream != NULL) */
    fga_aln_stream *stream;
    part_chain *chain;
    fga_session *Z;
    const fga_run_params *P;
    int64_t nlive, cover;
  } part_filter;

static void *part_filter_main(void *arg)
{ part_filter *F = arg;
  const double t0 = fga_wall();
  F->rc = fga_filter_alignments_mt(F->in,F->nthreads,&F->out);
  fga_note("part: redundancy filter",t0);
  if (F->rc == 0 && F->stream != NULL && F->P->reference_threads > 0)
    { const double t1 = fga_wall();
      F->rc = reference_order(F->Z,F->P,F->out);
      fga_note("part: the reference's tie order",t1);
    }
  if (F->rc != 0)
    F->err = strdup(fga_last_error());
  F->seconds = fga_wall() - t0;
  if (F->stream != NULL)
    { part_chain *C = F->chain;
      int go;
      pthread_mutex_lock(&C->mu);
      while (C->turn != F->part)
        pthread_cond_wait(&C->cv,&C->mu);
      if (F->rc != 0) C->failed = 1;
      go = !C->failed;
      pthread_mutex_unlock(&C->mu);
      if (go)
        { const double t1 = fga_wall();
          int64_t i;
          fga_aln_writer_threads(F->nthreads > 8 ? F->nthreads : 8);
          F->nlive = F->out->naln;
          for (i = 0; i < F->out->naln; i++)
            F->cover += F->out->alns[i].aepos - F->out->alns[i].abpos;
          if (fga_aln_stream_append(F->stream,F->out))
            { F->rc = 1;
              F->err = strdup(fga_last_error());
            }
          F->write_s = fga_wall() - t1;
          fga_note("part: records formatted and written",t1);
        }
      pthread_mutex_lock(&C->mu);
      if (F->rc != 0) C->failed = 1;
      C->turn = F->part + 1;
      pthread_cond_broadcast(&C->cv);
      pthread_mutex_unlock(&C->mu);
    }
  return NULL;
}

void fga_session_close(fga_session *Z)
{ if (Z == NULL) return;
  if (Z->dg2 != Z->dg1) fga_dgenome_free(Z->dg2);
  fga_dgenome_free(Z->dg1);
  fga_dgix_free(Z->d2); fga_dgix_free(Z->d1);
  fga_dev_close(Z->dev);
  if (!Z->borrowed_gix) { fga_gix_close(Z->x2); fga_gix_close(Z->x1); }
  if (!Z->borrowed_gdb) { fga_gdb_close(Z->g2); fga_gdb_close(Z->g1); }
  free(Z->cuts); free(Z->scount);
  free(Z);
}

int fga_gix_files_exist(const char *root)
{ char *p = NULL;
  size_t n = strlen(root);
  int ok;
  if (n > 4 && (strcmp(root+n-4,".gix") == 0 || strcmp(root+n-4,".gdb") == 0)) n -= 4;
  else if (n > 5 && strcmp(root+n-5,".1gdb") == 0) n -= 5;
  if (asprintf(&p,"%.*s.gix",(int) n,root) < 0) return 0;
  ok = access(p,R_OK) == 0;
  free(p);
  return ok;
}

/* load GDB + GIX of both genomes and make them resident in HBM */
int fga_session_open(const char *root1, const char *root2, int device, fga_session **out)
{ return fga_session_open_threads(root1,root2,device,8,out); }

/* nthreads: GIXmake's -T for an index the session has to build itself -- it decides the contig padding of a short GDB
   and the table parts (SURVEY.md hard part 9), i.e. the layout FastGA -T<n> would have got from its GIXmake call */

int fga_session_open_threads(const char *root1, const char *root2, int device, int nthreads, fga_session **out)
{ return fga_session_open_impl(root1,root2,device,nthreads,0,1,0,NULL,NULL,out); }

/* flags: FGA_SESSION_BUILD_INDEX -- the genome indices are built on the device even when <root>.gix files exist (they may
   be another program's: a parity run against the reference's own GIXmake output) */
int fga_session_open_flags(const char *root1, const char *root2, int device, int nthreads, int flags, fga_session **out)
{ return fga_session_open_impl(root1,root2,device,nthreads,0,1,flags,NULL,NULL,out); }

int fga_session_open_masked(const char *root1, const char *root2, int device, int nthreads, int flags,
                            const char *const *masks1, int nmasks1, const char *const *masks2, int nmasks2, fga_session **out)
{ fga_mask_args M;
  M.m1 = masks1; M.n1 = nmasks1; M.m2 = masks2; M.n2 = nmasks2;
  return fga_session_open_impl(root1,root2,device,nthreads,0,1,flags,&M,NULL,out);
}

int fga_session_open_sliced(const char *root1, const char *root2, int device, int nthreads, int rank, int nranks,
                            fga_session **out)
{ if (nranks < 1 || rank < 0 || rank >= nranks || nranks > 4096)
    { fga_set_error("fga_session_open_sliced: rank %d of %d",rank,nranks);
      *out = NULL;
      return 1;
    }
  return fga_session_open_impl(root1,root2,device,nthreads,rank,nranks,0,NULL,NULL,out);
}

/* prefix ranges of equal merge cost (entries of both tables + 2 per prefix) from the tables' per-prefix entry counts:
   cumulative counts on the host (int64 index arrays of the files), or count arrays of a device scan */
static void cuts_from_counts(const int64_t *idx1, const int64_t *idx2, const uint32_t *cnt1, const uint32_t *cnt2,
                             int nranks, int64_t *cuts)
{ int64_t total = 0, run = 0, p;
  int w = 1;
  if (idx1 != NULL)
    total = idx1[FGA_NPREFIX-1] + (idx2 != NULL ? idx2[FGA_NPREFIX-1] : idx1[FGA_NPREFIX-1]);
  else
    for (p = 0; p < FGA_NPREFIX; p++)
      total += (int64_t) cnt1[p] + (cnt2 != NULL ? cnt2[p] : cnt1[p]);
  total += 2*(int64_t) FGA_NPREFIX;
  cuts[0] = 0;
  for (p = 0; p < FGA_NPREFIX && w < nranks; p++)
    { int64_t c;
      if (idx1 != NULL)
        c = idx1[p] + (idx2 != NULL ? idx2[p] : idx1[p]) + 2*(p+1);
      else
        { run += (int64_t) cnt1[p] + (cnt2 != NULL ? cnt2[p] : cnt1[p]);
          c = run + 2*(p+1);
        }
      while (w < nranks && c > (total / nranks) * w)      /* smallest p whose inclusive cost exceeds the target */
        cuts[w++] = p < 1 ? 1 : p;
    }
  while (w <= nranks)
    cuts[w++] = FGA_NPREFIX;
  for (w = 1; w <= nranks; w++)
    if (cuts[w] < cuts[w-1]) cuts[w] = cuts[w-1];
}

int fga_session_open_impl(const char *root1, const char *root2, int device, int nthreads, int rank, int nranks, int flags,
                          const fga_mask_args *masks, const fga_shared_inputs *shared, fga_session **out)
{ fga_session *Z = calloc(1,sizeof(fga_session));
  void *img1 = NULL, *img2 = NULL;         /* the genomes' bases an index build left on the device */
  double t0;
  *out = NULL;
  if (Z == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  Z->self = (root2 == NULL);
  t0 = fga_wall();
  /* an index file is loaded when it is there; otherwise the index is built on the device from the GDB, straight
     into HBM (no .gix/.ktab files appear, like the reference without -k) */
  { const int build = (flags & FGA_SESSION_BUILD_INDEX) != 0;
    int have1 = !build && fga_gix_files_exist(root1), have2 = Z->self ? 1 : (!build && fga_gix_files_exist(root2));
    /* a genome with masks named gets its index built anew with their union as its soft mask (the reference runs GIXmake
       with the masks, FastGA.c:4739-4776) */
    if (masks != NULL && masks->n1 > 0) have1 = 0;
    if (masks != NULL && masks->n2 > 0 && !Z->self) have2 = 0;
    if (shared != NULL && shared->g1 != NULL)           /* opened (and masked) once by the caller for all of its ranks */
      { Z->borrowed_gdb = 1;
        Z->g1 = shared->g1; Z->g2 = Z->self ? NULL : shared->g2;
        if (have1 && (Z->self || have2) && shared->x1 != NULL && (Z->self || shared->x2 != NULL))
          { Z->borrowed_gix = 1;
            Z->x1 = shared->x1; Z->x2 = Z->self ? NULL : shared->x2;
          }
        else if ((have1 && fga_gix_open(root1,&Z->x1)) || (!Z->self && have2 && fga_gix_open(root2,&Z->x2)))
          goto fail;
      }
    else
      { if (fga_gdb_open(root1,&Z->g1) || (masks != NULL && masks->n1 > 0 && fga_gdb_apply_masks(Z->g1,masks->m1,masks->n1)) ||
            (have1 && fga_gix_open(root1,&Z->x1))) goto fail;
        if (!Z->self)
          { if (fga_gdb_open(root2,&Z->g2) || (masks != NULL && masks->n2 > 0 && fga_gdb_apply_masks(Z->g2,masks->m2,masks->n2)) ||
                (have2 && fga_gix_open(root2,&Z->x2))) goto fail;
          }
      }
    Z->load_s = fga_wall() - t0;
    if (fga_dev_open(device,&Z->dev)) goto fail;
    t0 = fga_wall();
    Z->devbuilt = !have1 || !have2;
    if (nthreads < 1) nthreads = 1;
    Z->nranks = nranks; Z->rank = rank;
    if (nranks <= 1)
      { if (have1 ? fga_dgix_upload(Z->dev,Z->x1,&Z->d1)
                  : fga_dgix_build_keep(Z->dev,Z->g1,nthreads,FGA_GIX_SOFT_MASK,0,FGA_NPREFIX,&Z->d1,&Z->x1,&img1)) goto fail;
        if (!Z->self && (have2 ? fga_dgix_upload(Z->dev,Z->x2,&Z->d2)
                               : fga_dgix_build_keep(Z->dev,Z->g2,nthreads,FGA_GIX_SOFT_MASK,0,FGA_NPREFIX,&Z->d2,&Z->x2,&img2)))
          goto fail;
      }
    else
      { /* the ranks' prefix ranges from the per-prefix entry counts of both tables (file indices, or one syncmer scan per
           genome on the device), then this rank's slice of each table: 1/N of the upload or of the build's sort */
        uint32_t *c1 = NULL, *c2 = NULL;
        int bad = 0;
        int64_t pb, pe;
        Z->cuts = malloc(sizeof(int64_t)*(nranks+1));
        if (Z->cuts == NULL) { fga_set_error("out of memory"); goto fail; }
        if (!have1)
          { c1 = malloc(sizeof(uint32_t)*FGA_NPREFIX);
            if (c1 == NULL || fga_dgix_prefix_counts(Z->dev,Z->g1,nthreads,c1)) bad = 1;
          }
        if (!bad && !Z->self && !have2)
          { c2 = malloc(sizeof(uint32_t)*FGA_NPREFIX);
            if (c2 == NULL || fga_dgix_prefix_counts(Z->dev,Z->g2,nthreads,c2)) bad = 1;
          }
        if (!bad && (have1 != 0) != (Z->self ? have1 != 0 : have2 != 0))
          { fga_set_error("a sliced session wants both genome indices as files, or neither");
            bad = 1;
          }
        if (bad) { free(c1); free(c2); goto fail; }
        cuts_from_counts(have1 ? Z->x1->index : NULL,(have1 && !Z->self) ? Z->x2->index : NULL,c1,Z->self ? NULL : c2,
                         nranks,Z->cuts);
        free(c1); free(c2);
        pb = Z->cuts[rank]; pe = Z->cuts[rank+1];
        if (pe <= pb) pe = pb + 1 <= FGA_NPREFIX ? pb + 1 : pb;           /* an empty range still needs a (tiny) table */
        if (pe <= pb) { pb = FGA_NPREFIX-1; pe = FGA_NPREFIX; }
        if (have1 ? fga_dgix_upload_range(Z->dev,Z->x1,pb,pe,&Z->d1)
                  : fga_dgix_build_keep(Z->dev,Z->g1,nthreads,FGA_GIX_SOFT_MASK,pb,pe,&Z->d1,&Z->x1,&img1)) goto fail;
        if (!Z->self && (have2 ? fga_dgix_upload_range(Z->dev,Z->x2,pb,pe,&Z->d2)
                               : fga_dgix_build_keep(Z->dev,Z->g2,nthreads,FGA_GIX_SOFT_MASK,pb,pe,&Z->d2,&Z->x2,&img2)))
          goto fail;
      }
    /* the builder's key buffers (2 x 16 B per k-mer) stay in their workspace slots: the comparison's first large
       buffers (seeds, per-part staging) take them over (fga_dev_acquire) */
  }
  if (!Z->self && (Z->x1->legacy != 0) != (Z->x2->legacy != 0))      /* the reference refuses the mix too (FastGA.c:4882-4886) */
    { fga_set_error("one genome index is in the pre-v1.3 layout and the other is not: rebuild the old one");
      goto fail;
    }
  if (Z->x1->nctg < Z->g1->ncontig || (!Z->self && Z->x2->nctg < Z->g2->ncontig))
    { fga_set_error("genome index and genome database disagree on the number of contigs");
      goto fail;
    }
  /* the bases an index build left on the device are taken over (the call owns the image from here on, also when it fails) */
  { void *im = img1;
    img1 = NULL;
    if (im != NULL ? fga_dgenome_adopt(Z->dev,Z->g1,Z->x1->perm,Z->x1->nctg,1,im,&Z->dg1)
                   : fga_dgenome_upload(Z->dev,Z->g1,Z->x1->perm,Z->x1->nctg,1,&Z->dg1)) goto fail;
  }
  if (Z->self)
    Z->dg2 = Z->dg1;
  else
    { void *im = img2;
      img2 = NULL;
      if (im != NULL ? fga_dgenome_adopt(Z->dev,Z->g2,Z->x2->perm,Z->x2->nctg,1,im,&Z->dg2)
                     : fga_dgenome_upload(Z->dev,Z->g2,Z->x2->perm,Z->x2->nctg,1,&Z->dg2)) goto fail;
    }
  Z->upload_s = fga_wall() - t0;
  *out = Z;
  return 0;
fail:
  if (img1 != NULL && Z != NULL && Z->dev != NULL) fga_dev_free(Z->dev,img1);
  if (img2 != NULL && Z != NULL && Z->dev != NULL) fga_dev_free(Z->dev,img2);
  fga_session_close(Z);
  return 1;
}

fga_dev *fga_session_device(fga_session *Z) { return Z->dev; }
int64_t  fga_session_table_bytes(const fga_session *Z)
{ return fga_dgix_nents(Z->d1)*Z->x1->ebytes + (Z->self ? 0 : fga_dgix_nents(Z->d2)*Z->x2->ebytes); }
int      fga_session_seed_bytes(const fga_session *Z)
{ return 1 + Z->x1->postbytes + Z->x1->contbytes + (Z->self ? Z->x1->postbytes + Z->x1->contbytes
                                                            : Z->x2->postbytes + Z->x2->contbytes); }
int64_t  fga_session_bases(const fga_session *Z, int which)
{ return which == 0 ? Z->g1->seqtot : (Z->self ? Z->g1->seqtot : Z->g2->seqtot); }

/* ---- the hot path in three stages, so that one comparison can be cut at the two places where the reference itself
 *      re-partitions its data: between phase 1 and phase 2 (seeds are regrouped from k-mer prefix ranges to A-contig
 *      parts, FastGA.c:5097-5134, 5160-5184) and before phase 3 (the records of all parts are merged, FastGA.c:3991-4133).
 *      One GPU runs them back to back (fga_session_run); N GPUs run merge on their prefix range, exchange seeds by part
 *      (fga_seeds_split_to / RCCL all-to-all-v / fga_seeds_import), run align on their part, and rank 0 runs finish on
 *      the gathered records (bench.py, fastga_amd/parallel.py). ------------------------------------------------------- */

/* phase 1 over the 12-mer prefix range [prefix_begin, prefix_end) (0,0 = all): adaptive seeds in HBM */
int fga_session_merge(fga_session *Z, const fga_run_params *P, int64_t prefix_begin, int64_t prefix_end,
                      fga_dseeds **out, fga_run_stats *S)
{ fga_gix *x1 = Z->x1, *x2 = Z->x2;
  fga_dev *dev = Z->dev;
  const int self = Z->self;
  fga_dseeds *seeds = NULL;
  fga_merge_params mp;
  double t0 = fga_wall();
  int rc;

  *out = NULL;
  { const fga_gix *xs[2] = { x1, self ? NULL : x2 };      /* an index in the old layout has a cutoff built in (FastGA.c:4959-4974) */
    int q;
    for (q = 0; q < 2; q++)
      if (xs[q] != NULL && xs[q]->legacy && xs[q]->freq < P->freq)
        { fga_set_error("genome index %d was built with a frequency cutoff of %d < the requested cutoff %d",q+1,xs[q]->freq,P->freq);
          return 1;
        }
  }
  if (Z->nranks > 1)                     /* a sliced session can only merge (part of) its own prefix range */
    { const int64_t lo = Z->cuts[Z->rank], hi = Z->cuts[Z->rank+1];
      if (prefix_begin == 0 && prefix_end == 0)
        { if (hi > lo)
            { prefix_begin = lo; prefix_end = hi; }
          else                                  /* an empty range of its own: the table is a placeholder slice of another */
            prefix_begin = prefix_end = lo >= 1 ? lo : 1;      /* rank's prefixes; (0,0) would read as "everything" */
        }
      if (!(prefix_begin == prefix_end) && (prefix_begin < lo || prefix_end > hi))
        { fga_set_error("the session holds the 12-mer prefixes [%lld,%lld) of the tables, the merge asks for [%lld,%lld)",
                        (long long) lo,(long long) hi,(long long) prefix_begin,(long long) prefix_end);
          return 1;
        }
    }
  memset(&mp,0,sizeof(mp));
  mp.freq = P->freq; mp.soft_mask = P->soft_mask; mp.flip = 0;
  mp.prefix_begin = prefix_begin; mp.prefix_end = prefix_end;
  /* first guess of the buffer: two seeds per table-1 entry of the range, one per entry beyond 2^28 entries (the merge
     reports the exact need when that is not enough and is repeated once -- tens of milliseconds at human scale, where
     the large guess would be a fresh allocation of 77 GB instead of a take-over of the index builder's key buffer),
     never more than a quarter of the device memory still to be had */
  { int64_t guess = (P->symmetric && !self) ? 2*(x1->nents + x2->nents) + (1<<20) : 0;
    if (Z->nranks <= 1 && prefix_begin == 0 && (prefix_end <= 0 || prefix_end >= FGA_NPREFIX))
      { const int64_t room = (int64_t) (fga_dev_available(dev) / 4 / sizeof(fga_seed));
        if (guess == 0)
          guess = (x1->nents > ((int64_t) 1 << 28) ? x1->nents : 2*x1->nents) + (1<<20);      /* one per entry at human scale */
        if (guess > room && room > x1->nents/2)
          guess = room;
      }
    rc = fga_seed_merge(dev,Z->d1,self ? NULL : Z->d2,&mp,guess,&seeds);
  }
  if (rc == 2)
    { int64_t need = fga_seeds_count(seeds) + 1024;
      if (P->symmetric && !self) need += x2->nents + x1->nents;
      fga_seeds_free(seeds); seeds = NULL;
      rc = fga_seed_merge(dev,Z->d1,self ? NULL : Z->d2,&mp,need,&seeds);
    }
  if (rc) goto fail;
  if (S != NULL) S->merge_kernel_ms += fga_dev_stage_ms(dev,FGA_STAGE_MERGE);
  if (P->symmetric && !self)
    { mp.flip = 1;
      rc = fga_seed_merge_append(dev,Z->d2,Z->d1,&mp,seeds);
      if (rc) goto fail;
      if (S != NULL) S->merge_kernel_ms += fga_dev_stage_ms(dev,FGA_STAGE_MERGE);
    }
  if (P->reference_threads > 0)          /* the reference's buck[] per stream: what its search threads' ranges are cut from */
    { const int nctg = x1->nctg;
      int64_t *c = malloc(sizeof(int64_t)*2*nctg);
      int j;
      if (c == NULL || (Z->scount == NULL && (Z->scount = calloc(2*(size_t) nctg,sizeof(int64_t))) == NULL))
        { free(c); fga_set_error("out of memory"); goto fail; }
      if (fga_seeds_strand_histogram(dev,seeds,nctg,c)) { free(c); goto fail; }
      for (j = 0; j < 2*nctg; j++) Z->scount[j] += c[j];
      free(c);
    }
  if (S != NULL)
    { int64_t n = fga_seeds_count(seeds), l = fga_seeds_plen_sum(seeds);
      if (self) { n /= 2; l /= 2; }
      S->nseeds += n; S->seed_len_sum += l;
      S->merge_s += fga_wall() - t0;
    }
  *out = seeds;
  return 0;
fail:
  fga_seeds_free(seeds);
  return 1;
}

/* phase 2 on a set of seeds (all of them, or one A-contig part): diagonal records, sort, chain scan, wave extension.
   The seed buffer is consumed.  *raw: the accepted alignments in discovery order (unit, seq), before the redundancy
   filter, which needs all records of a contig pair -- they all come from the part that owns the A contig. */
int fga_session_align(fga_session *Z, const fga_run_params *P, fga_dseeds *seeds, fga_alns **raw, fga_run_stats *S)
{ fga_gdb *g1 = Z->g1, *g2 = Z->g2;
  fga_gix *x1 = Z->x1, *x2 = Z->x2;
  fga_dev *dev = Z->dev;
  fga_dkeys *keys = NULL;
  fga_hits *hits = NULL;
  int64_t *alen = NULL;
  int16_t *table = NULL;
  const int self = Z->self;
  int status = 1, i;
  double t0, t1;
  fga_run_stats st;

  *raw = NULL;
  memset(&st,0,sizeof(st));
  fga_dev_set_host_threads(dev,P->nthreads);
  t0 = fga_wall();
  { fga_sort_params sp;
    sp.amxpos = g1->maxctg; sp.bmxpos = self ? g1->maxctg : g2->maxctg;
    sp.nctg_a = x1->nctg;   sp.nctg_b = self ? x1->nctg : x2->nctg;
    sp.anti_order_only = 1;          /* chains do not depend on the order inside an (anti-diagonal) tie */
    if (fga_seed_sort(dev,seeds,&sp,&keys)) goto done;
    fga_dev_peak_bytes(dev);                 /* seeds + both key buffers are live here: the footprint's peak */
    fga_seeds_free(seeds); seeds = NULL;
    st.sort_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_SORT);
    { int wa, wb, wd, wt;
      fga_keys_layout(keys,&wa,&wb,&wd,&wt);
      st.sort_keys = fga_keys_count(keys);
      st.sort_passes = (1 + wa + wb + wd + wt + 7) / 8;        /* the bits above diag&63 | lcp (anti_order_only) */
    }
  }
  t1 = fga_wall();
  st.sort_s = t1 - t0;

  { fga_chain_params cp;
    alen = malloc(sizeof(int64_t)*x1->nctg);
    if (alen == NULL)
      { fga_set_error("out of memory");
        goto done;
      }
    for (i = 0; i < x1->nctg; i++)
      alen[i] = (x1->perm[i] < g1->ncontig) ? g1->contigs[x1->perm[i]].clen : FGA_KMER;
    cp.chain_break = P->chain_break; cp.chain_min = P->chain_min;
    cp.amxpos = g1->maxctg; cp.bmxpos = self ? g1->maxctg : g2->maxctg;
    cp.alen = alen; cp.nalen = x1->nctg;
    if (fga_chain_scan_device(dev,keys,&cp,&hits)) goto done;      /* the sorted records never leave HBM */
    fga_keys_free(keys); keys = NULL;
    st.nhits = hits->nhits;
    st.nunits = hits->nunits;
    st.chain_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_CHAIN);
    st.chain_s = fga_wall() - t1;
  }

  t1 = fga_wall();
  { fga_extend_params ep;
    int path_ave;
    memset(&ep,0,sizeof(ep));
    table = malloc(sizeof(int16_t)*2*32768);
    if (table == NULL)
      { fga_set_error("out of memory");
        goto done;
      }
    fga_align_spec(1.-P->align_rate,100,g1->freq,&path_ave,table,table+32768);
    ep.tspace = 100; ep.path_ave = path_ave; ep.table = table; ep.score = table+32768;
    ep.self = self; ep.aln_min = P->align_min - 50; ep.aln_rate = P->align_rate + .05;
    if (fga_extend(dev,Z->dg1,Z->dg2,hits,&ep,raw)) goto done;
    st.nalns = (*raw)->naln; st.ncalls = (*raw)->ncalls; st.nwaves = (*raw)->nwaves;
    st.ext_cells = (*raw)->ncells; st.ext_bases = (*raw)->nbases; st.ext_trace = (*raw)->ntrace;
    st.ext_busy_waves = (*raw)->busy_waves;
    st.extend_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_EXTEND);
  }
  st.extend_s = fga_wall() - t1;
  status = 0;

done:
  if (S != NULL)
    { S->sort_s += st.sort_s; S->chain_s += st.chain_s; S->extend_s += st.extend_s;
      S->sort_kernel_ms += st.sort_kernel_ms; S->extend_kernel_ms += st.extend_kernel_ms; S->chain_kernel_ms += st.chain_kernel_ms;
      S->nhits += st.nhits; S->nunits += st.nunits; S->nalns += st.nalns; S->ncalls += st.ncalls; S->nwaves += st.nwaves;
      S->ext_cells += st.ext_cells; S->ext_bases += st.ext_bases; S->ext_trace += st.ext_trace;
      S->sort_keys += st.sort_keys;
      if (st.sort_passes > S->sort_passes) S->sort_passes = st.sort_passes;
      if (st.ext_busy_waves > 0.) S->ext_busy_waves = st.ext_busy_waves;
    }
  free(alen); free(table);
  fga_hits_free(hits);
  fga_keys_free(keys);
  fga_seeds_free(seeds);
  return status;
}

/* the records of several parts as one set, unit numbers made distinct (discovery order is (part, unit, seq)) */
int fga_alns_concat(const fga_alns *const *raw, int nraw, fga_alns **out)
{ fga_alns *R = calloc(1,sizeof(fga_alns));
  int64_t at = 0, tat = 0;
  int32_t ubase = 0;
  int k;
  *out = NULL;
  if (R == NULL) goto oom;
  for (k = 0; k < nraw; k++)
    if (raw[k] != NULL)
      { R->naln += raw[k]->naln; R->ntrace += raw[k]->ntrace; R->ncalls += raw[k]->ncalls; R->nwaves += raw[k]->nwaves;
        R->ncells += raw[k]->ncells; R->nbases += raw[k]->nbases;        /* the launches' accounting adds up; busy wavefronts: the largest */
        if (raw[k]->busy_waves > R->busy_waves) R->busy_waves = raw[k]->busy_waves;
      }
  R->alns = malloc(sizeof(fga_aln)*(R->naln+1));
  R->tbytes = malloc(R->ntrace+16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;
  for (k = 0; k < nraw; k++)
    if (raw[k] != NULL)
      { int64_t i;
        int32_t umax = -1;
        for (i = 0; i < raw[k]->naln; i++)
          { fga_aln a = raw[k]->alns[i];
            if (a.unit > umax) umax = a.unit;
            if (a.unit >= 0) a.unit += ubase;
            a.toff += tat;
            R->alns[at++] = a;
          }
        if (raw[k]->ntrace > 0)
          memcpy(R->tbytes+tat,raw[k]->tbytes,raw[k]->ntrace);
        tat += raw[k]->ntrace;
        ubase += umax+1;
      }
  *out = R;
  return 0;
oom:
  fga_set_error("out of memory");
  if (R != NULL) { free(R->alns); free(R->tbytes); free(R); }
  return 1;
}

/* phase 3 for a set in final order: .1aln, PAF / PSL */
/* records that tie on (aread, abpos) in the order FastGA -T<reference_threads> writes them (fga_order.c) */
static int reference_order(fga_session *Z, const fga_run_params *P, fga_alns *fin)
{ const fga_gdb *g1 = Z->g1, *g2 = Z->self ? Z->g1 : Z->g2;
  const fga_gix *x1 = Z->x1, *x2 = Z->self ? Z->x1 : Z->x2;
  const int T = P->reference_threads;
  int nc = x1->nctg, j, rc = 1, swide, dbyte = 0;
  int64_t *clen = NULL, *cnt = NULL, amx = 0, bmx = 0, cum = 1;
  int *slot = NULL, *invp = NULL;
  if (Z->scount == NULL)
    { fga_set_error("reference order wanted, but no merge of this session counted its seeds per strand (reference_threads "
                    "must be set for fga_session_merge too, or fga_session_set_strand_counts called)");
      return 1;
    }
  if (nc < T) nc = T;                                  /* short_GDB_fix: at least one contig per thread */
  if (nc < g1->ncontig) nc = g1->ncontig;
  clen = malloc(sizeof(int64_t)*nc); cnt = calloc(2*(size_t) nc,sizeof(int64_t));
  slot = malloc(sizeof(int)*2*nc); invp = malloc(sizeof(int)*nc);
  if (clen == NULL || cnt == NULL || slot == NULL || invp == NULL)
    { fga_set_error("out of memory"); goto done; }
  for (j = 0; j < nc; j++)
    { clen[j] = (j < x1->nctg && x1->perm[j] < g1->ncontig) ? g1->contigs[x1->perm[j]].clen : FGA_KMER;
      invp[j] = 0;
      if (clen[j] > amx) amx = clen[j];
    }
  for (j = 0; j < x1->nctg; j++)
    { if (x1->perm[j] < g1->ncontig) invp[x1->perm[j]] = j;
      cnt[j] = Z->scount[j]; cnt[nc + j] = Z->scount[x1->nctg + j];
    }
  bmx = g2->maxctg;
  if (x2->nctg > g2->ncontig && bmx < FGA_KMER) bmx = FGA_KMER;
  if (Z->self) bmx = amx;
  while (cum < amx + bmx) { cum *= 256; dbyte += 1; }              /* DBYTE (FastGA.c:5040-5046) */
  swide = 2*dbyte + x2->contbytes + 2;                             /* FastGA.c:4190 */
  if (fga_reference_slots(cnt,clen,nc,T,swide,slot)) goto done;
  rc = fga_alns_reference_order(fin,slot,invp,nc);
done:
  free(clen); free(cnt); free(slot); free(invp);
  return rc;
}

static int finish_output(fga_session *Z, const fga_run_params *P, fga_alns *fin, fga_run_stats *st)
{ fga_gdb *g1 = Z->g1, *g2 = Z->g2;
  fga_dev *dev = Z->dev;
  const int self = Z->self;
  int64_t i;
  double t1;

  t1 = fga_wall();
  if (P->reference_threads > 0 && reference_order(Z,P,fin))
    return 1;
  fga_note("finish: the reference's tie order",t1);
  st->nlive = fin->naln;
  for (i = 0; i < fin->naln; i++)
    st->cover += fin->alns[i].aepos - fin->alns[i].abpos;
  t1 = fga_wall();
  if (P->out_path != NULL)
    { char *n1 = NULL, *n2 = NULL;
      const char *cmd = P->command_line ? P->command_line : "FastGA";
      int rc;
      if (asprintf(&n1,"%s",g1->path) < 0) n1 = NULL;
      if (!self && asprintf(&n2,"%s",g2->path) < 0) n2 = NULL;
      fga_aln_writer_threads(P->nthreads > 8 ? P->nthreads : 8);
      /* the binary ONEcode container, like the reference; FGA_ALN_ASCII=1 selects the text form */
      if (getenv("FGA_ALN_ASCII") != NULL && atoi(getenv("FGA_ALN_ASCII")) != 0)
        rc = fga_write_1aln(P->out_path,g1,self ? NULL : g2,fin,100,n1 ? n1 : "genome1",n2,cmd);
      else
        rc = fga_write_1aln_binary(P->out_path,g1,self ? NULL : g2,fin,100,n1 ? n1 : "genome1",n2,cmd);
      free(n1); free(n2);
      if (rc) return 1;
    }
  st->write_s = fga_wall() - t1;
  fga_note("finish: .1aln written",t1);

  /* ---- PAF (what the reference leaves to a second process, ALNtoPAF): outside the .1aln clock ---- */
  if (P->paf_path != NULL)
    { fga_traces *tr = NULL;
      const int psl = (P->paf_flags & FGA_OUT_PSL) != 0;
      const int bases = psl || (P->paf_flags & (FGA_PAF_CIGAR_M|FGA_PAF_CIGAR_X|FGA_PAF_CS_SHORT|FGA_PAF_CS_LONG)) != 0;
      int rc = 0;
      t1 = fga_wall();
      if (bases)
        { rc = fga_trace_pts_regrouped(dev,Z->dg1,Z->dg2,fin,100,0,&tr);   /* Compute_Trace_PTS + Gap_Improver */
          st->trace_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_TRACE) + fga_dev_stage_ms(dev,FGA_STAGE_REGROUP);
        }
      st->trace_s = fga_wall() - t1;
      t1 = fga_wall();
      if (rc == 0)
        rc = psl ? fga_write_psl(P->paf_path,g1,self ? NULL : g2,fin,tr,P->nthreads)
                 : fga_write_paf(P->paf_path,g1,self ? NULL : g2,fin,tr,P->paf_flags,P->nthreads);
      st->paf_s = fga_wall() - t1;
      fga_traces_free(tr);
      if (rc) return 1;
    }
  return 0;
}

static void finish_stats(fga_run_stats *S, const fga_run_stats *st)
{ if (S == NULL) return;
  S->nlive = st->nlive; S->cover = st->cover; S->filter_s += st->filter_s; S->write_s += st->write_s;
  S->trace_s = st->trace_s; S->paf_s = st->paf_s; S->trace_kernel_ms = st->trace_kernel_ms;
}

/* redundancy filter + phase 3 over the records of all parts: order, .1aln, PAF / PSL */
int fga_session_finish(fga_session *Z, const fga_run_params *P, const fga_alns *const *raw, int nraw, fga_run_stats *S)
{ fga_alns *all = NULL, *fin = NULL;
  const fga_alns *in;
  int status = 1;
  double t1;
  fga_run_stats st;

  memset(&st,0,sizeof(st));
  t1 = fga_wall();
  if (nraw == 1 && raw[0] != NULL)
    in = raw[0];
  else
    { if (fga_alns_concat(raw,nraw,&all)) goto done;
      in = all;
    }
  if (fga_filter_alignments_mt(in,P->nthreads,&fin)) goto done;
  st.filter_s = fga_wall() - t1;
  if (finish_output(Z,P,fin,&st)) goto done;
  status = 0;

done:
  finish_stats(S,&st);
  fga_alns_free(all); fga_alns_free(fin);
  return status;
}

/* the same for record sets that went through the redundancy filter already, part by part (every contig pair's records
   come from one part): their runs per A contig are laid out in order (fga_alns_merge_filtered) and written */
int fga_session_finish_filtered(fga_session *Z, const fga_run_params *P, const fga_alns *const *filtered, int nsets,
                                fga_run_stats *S)
{ fga_alns *fin = NULL;
  int status = 1;
  double t1;
  fga_run_stats st;

  memset(&st,0,sizeof(st));
  t1 = fga_wall();
  if (fga_alns_merge_filtered_mt(filtered,nsets,P->nthreads,&fin)) goto done;
  st.filter_s = fga_wall() - t1;
  if (finish_output(Z,P,fin,&st)) goto done;
  status = 0;

done:
  finish_stats(S,&st);
  fga_alns_free(fin);
  return status;
}

/* A-contig parts of the seeds of this session (weights: seeds per A contig, summed over ranks by the caller when there
   are several): see fga_partition_contigs */
int fga_session_nctg(const fga_session *Z) { return Z->x1->nctg; }
const int *fga_session_contig_perm(const fga_session *Z) { return Z->x1->perm; }

int fga_session_strand_counts(const fga_session *Z, int64_t *counts)
{ if (Z == NULL || counts == NULL)
    { fga_set_error("fga_session_strand_counts: null argument");
      return 1;
    }
  if (Z->scount == NULL)
    memset(counts,0,sizeof(int64_t)*2*Z->x1->nctg);
  else
    memcpy(counts,Z->scount,sizeof(int64_t)*2*Z->x1->nctg);
  return 0;
}

int fga_session_set_strand_counts(fga_session *Z, const int64_t *counts)
{ if (Z == NULL || counts == NULL)
    { fga_set_error("fga_session_set_strand_counts: null argument");
      return 1;
    }
  if (Z->scount == NULL && (Z->scount = malloc(sizeof(int64_t)*2*Z->x1->nctg)) == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  memcpy(Z->scount,counts,sizeof(int64_t)*2*Z->x1->nctg);
  return 0;
}

void fga_session_clear_strand_counts(fga_session *Z)       /* before the merges of a new comparison on the same session */
{ if (Z != NULL && Z->scount != NULL)
    memset(Z->scount,0,sizeof(int64_t)*2*Z->x1->nctg);
}

int fga_session_prefix_cuts(fga_session *Z, int nshards, int64_t *cuts)
{ if (Z->nranks > 1)                     /* a sliced session: the ranges it was opened with */
    { if (nshards != Z->nranks)
        { fga_set_error("the session holds rank %d's slice of %d prefix ranges, %d were asked for",Z->rank,Z->nranks,nshards);
          return 1;
        }
      memcpy(cuts,Z->cuts,sizeof(int64_t)*(nshards+1));
      return 0;
    }
  return fga_merge_prefix_cuts(Z->dev,Z->d1,Z->self ? NULL : Z->d2,nshards,cuts);
}

/* one pass of the hot path over the resident inputs: phases 1-3.  When the merge finds more seeds than one sort pass
   should take (P->pass_seeds, default 1.5 G: the sort's tile counters are 32-bit and three 16-byte buffers per seed are
   live), phase 2 runs part by part over an A-contig partition -- the reference's own NPARTS loop (FastGA.c:5186-5204). */
int fga_session_run(fga_session *Z, const fga_run_params *P, fga_run_stats *S)
{ fga_dev *dev = Z->dev;
  fga_dseeds *seeds = NULL;
  fga_alns **raw = NULL;
  int nparts = 1, p, status = 1, finished = 0;
  part_filter *pf = NULL;
  part_chain chain;
  fga_aln_stream *stream = NULL;
  int streaming = 0, chain_made = 0;
  int64_t limit = P->pass_seeds > 0 ? P->pass_seeds : (int64_t) 1500000000;
  int64_t *cnt = NULL, *poff = NULL;
  int *select = NULL;
  void *stage = NULL;
  double tstart = fga_wall();
  fga_run_stats st;

  memset(&st,0,sizeof(st));
  st.load_s = Z->load_s; st.upload_s = Z->upload_s;
  fga_session_clear_strand_counts(Z);
  if (fga_session_merge(Z,P,0,0,&seeds,&st)) goto done;
  fga_note("run: seed merge (incl. buffers)",tstart);
  { int64_t n = fga_seeds_count(seeds);
    if (n > limit)
      nparts = (int) ((n + limit - 1) / limit);
    /* FGA_OVERLAP_PARTS=<n>: cut a one-pass run into n parts all the same, so that the host work on one part's records
       (redundancy filter) runs beside the next part's kernels.  Off by default: at 10^6 contig pairs (150 Mbp repeat-heavy
       self comparison) four parts cost more in per-part tails (579 ms) than the overlap returns (536 ms in one pass) */
    if (nparts == 1 && getenv("FGA_OVERLAP_PARTS") != NULL && atoi(getenv("FGA_OVERLAP_PARTS")) > 1)
      nparts = atoi(getenv("FGA_OVERLAP_PARTS"));
    if (nparts > 64) nparts = 64;
    if (nparts > Z->x1->nctg) nparts = Z->x1->nctg;
  }
  raw = calloc(nparts,sizeof(fga_alns *));
  if (raw == NULL)
    { fga_set_error("out of memory");
      goto done;
    }
  if (nparts == 1)
    { if (fga_session_align(Z,P,seeds,&raw[0],&st)) { seeds = NULL; goto done; }
      seeds = NULL;
    }
  else
    { const int nctg = Z->x1->nctg;
      const int64_t n = fga_seeds_count(seeds);
      cnt = malloc(sizeof(int64_t)*nctg); poff = malloc(sizeof(int64_t)*(nparts+1)); select = malloc(sizeof(int)*nctg);
      if (cnt == NULL || poff == NULL || select == NULL)
        { fga_set_error("out of memory");
          goto done;
        }
      /* The .1aln as a stream: with the A contigs dealt to the parts in ORIGINAL order (the output's primary order) part p's
         records all come before part p+1's, so they are filtered, ordered, formatted and written while the next part's
         kernels run; the finish is the footer.  Not with PAF / PSL output (which wants the whole set) or the text form */
      streaming = fga_run_can_stream(P);
      /* seeds per A contig: the two strands' counts the merge has just taken for the reference's tie order, or a pass of its own */
      if (P->reference_threads > 0 && Z->scount != NULL)
        { int j;
          for (j = 0; j < nctg; j++)
            cnt[j] = Z->scount[j] + Z->scount[nctg + j];
        }
      else if (fga_seeds_contig_histogram(dev,seeds,nctg,cnt))
        goto done;
      if (streaming)
        { int64_t big = 0, *sum = calloc(nparts,sizeof(int64_t));
          int j;
          if (sum == NULL)
            { fga_set_error("out of memory");
              goto done;
            }
          /* (equal stretches: a smaller last pass would shorten the host work behind its kernels -- 1.27 -> 1.24 s on the
             3 Gbp pair at 1 % -- but a pass's kernel time has a floor, the serial chain of its longest alignment, and at 10 %
             the small pass sits on it: 2.84 -> 3.01 s) */
          if (fga_partition_contigs_in_order(cnt,Z->x1->perm,nctg,nparts,select))
            { free(sum);
              goto done;
            }
          for (j = 0; j < nctg; j++)
            if ((sum[select[j]] += cnt[j]) > big) big = sum[select[j]];
          free(sum);
          if (big > limit + limit/2 && big > n/nparts + n/(2*nparts) && !fga_run_stream_forced())     /* a stretch half again over a pass's seeds (one */
            streaming = 0;                                                /* contig dominates): the balanced deal instead  */
        }
      if (!streaming && fga_partition_contigs(cnt,nctg,nparts,select))
        goto done;
      if ((stage = fga_dev_stage_acquire(dev,(size_t) n*sizeof(fga_seed) + 64)) == NULL) goto done;
      if (fga_seeds_split_to(dev,seeds,select,nctg,nparts,stage,poff)) goto done;
      fga_seeds_free(seeds); seeds = NULL;   /* its slot is taken over by the parts' buffers */
      fga_note("run: seeds regrouped by A-contig part",tstart);
      pf = calloc(nparts,sizeof(part_filter));
      if (pf == NULL)
        { fga_set_error("out of memory");
          goto done;
        }
      if (streaming)
        { if (fga_session_stream_open(Z,P,&stream)) goto done;
          pthread_mutex_init(&chain.mu,NULL); pthread_cond_init(&chain.cv,NULL);
          chain.turn = 0; chain.failed = 0;
          chain_made = 1;
        }
      for (p = 0; p < nparts; p++)
        { const void *src = (const char *) stage + (size_t) poff[p]*sizeof(fga_seed);
          const int64_t c = poff[p+1] - poff[p];
          fga_dseeds *part = NULL;
          if (fga_seeds_view(dev,src,c,&part)) goto done;        /* (the piece itself: the sort's first pass reads it once) */
          if (fga_session_align(Z,P,part,&raw[p],&st)) goto done;
          fga_note("run: part aligned",tstart);
          /* this part's records through the redundancy filter in the background (every contig pair's records are in the
             part that owns the A contig) while the next part's kernels run */
          /* half of the run's threads beside the next part's kernels (whose host tails want the rest); all of them for the
             last part, which nothing runs beside */
          pf[p].in = raw[p]; pf[p].nthreads = p == nparts-1 ? P->nthreads : (P->nthreads > 2 ? P->nthreads/2 : 1);
          pf[p].part = p; pf[p].stream = stream; pf[p].chain = &chain; pf[p].Z = Z; pf[p].P = P;
          if (pthread_create(&pf[p].th,NULL,part_filter_main,&pf[p]) == 0)
            pf[p].started = 1;
          else
            part_filter_main(&pf[p]);
        }
      fga_dev_stage_release(dev,stage); stage = NULL;
      { const fga_alns **fsets = calloc(nparts,sizeof(fga_alns *));
        int bad = (fsets == NULL);
        const double tj = fga_wall();
        for (p = 0; p < nparts; p++)
          { if (pf[p].started) { pthread_join(pf[p].th,NULL); pf[p].started = 0; }
            if (pf[p].rc != 0 && !bad)
              { fga_set_error("%s",pf[p].err ? pf[p].err : "alignment filter failed");
                bad = 1;
              }
            if (fsets != NULL) fsets[p] = pf[p].out;
          }
        st.filter_s += fga_wall() - tj;                 /* what the filter (and a stream's last stretch) added to the critical path */
        fga_note(streaming ? "run: filters joined, parts written" : "run: filters joined",tstart);
        if (streaming)
          { const double tw = fga_wall();
            for (p = 0; p < nparts; p++)
              { st.nlive += pf[p].nlive; st.cover += pf[p].cover; st.write_s += pf[p].write_s; }
            if (fga_aln_stream_close(stream,!bad)) bad = 1;
            stream = NULL;
            st.streamed_parts = nparts;
            st.write_s += fga_wall() - tw;
            fga_note("finish: .1aln closed (footer)",tw);
          }
        else if (!bad && fga_session_finish_filtered(Z,P,fsets,nparts,&st)) bad = 1;
        free(fsets);
        if (bad) goto done;
      }
      finished = 1;
    }
  if (!finished && fga_session_finish(Z,P,(const fga_alns *const *) raw,nparts,&st)) goto done;
  st.phase23_s = fga_wall() - tstart - st.trace_s - st.paf_s;
  st.nparts = nparts;
  st.hbm_peak_bytes = fga_dev_peak_bytes(dev);
  st.bases1 = Z->g1->seqtot; st.bases2 = Z->self ? Z->g1->seqtot : Z->g2->seqtot;
  status = 0;

done:
  if (S != NULL) *S = st;
  if (pf != NULL)
    for (p = 0; p < nparts; p++)
      { if (pf[p].started) pthread_join(pf[p].th,NULL);
        fga_alns_free(pf[p].out); free(pf[p].err);
      }
  free(pf);
  if (stream != NULL) fga_aln_stream_close(stream,0);      /* (a failed run leaves no file) */
  if (chain_made) { pthread_mutex_destroy(&chain.mu); pthread_cond_destroy(&chain.cv); }
  if (raw != NULL)
    for (p = 0; p < nparts; p++) fga_alns_free(raw[p]);
  free(raw); free(cnt); free(poff); free(select);
  if (stage != NULL) fga_dev_stage_release(dev,stage);
  fga_seeds_free(seeds);
  return status;
}

int fga_run(const char *root1, const char *root2, const fga_run_params *P, fga_run_stats *S)
{ fga_session *Z;
  int rc;
  if (fga_session_open_masked(root1,root2,P->device,P->nthreads > 0 ? P->nthreads : 8,
                              P->build_index ? FGA_SESSION_BUILD_INDEX : 0,P->masks1,P->nmasks1,P->masks2,P->nmasks2,&Z))
    return 1;
  if ((P->nmasks1 > 0 || P->nmasks2 > 0) && !P->soft_mask)
    { fga_run_params Q = *P;            /* masks named: the comparison runs with soft masking on (FastGA.c:4580) */
      Q.soft_mask = 1;
      rc = fga_session_run(Z,&Q,S);
    }
  else
    rc = fga_session_run(Z,P,S);
  fga_session_close(Z);
  return rc;
}
