/* fga_filter.c -- redundancy removal and final ordering of the accepted alignments (host).
 *
 * Replaces the tail of align_contigs (reference FastGA.c:3405-3694: ALIGN_SORT, the two elimination
 * passes, entwine FastGA.c:2818-2941) and the ordering of la_sort / la_merge (FastGA.c:3800-3835, 3906-3918).
 *
 * Per (A contig, B contig, strand), over the alignments in discovery order (unit by unit, in hit order):
 *   1. stable sort by abpos;
 *   2. pass 1 (j descending, k > j while a-intervals overlap): identical start => drop the one whose a-interval
 *      ends first (identical boxes: the reference compares `diffs < aepos`, kept literally); identical end =>
 *      drop the one that starts later;
 *   3. pass 2: for surviving overlapping pairs whose b-intervals intersect, walk both traces trace point by trace
 *      point ("entwine"); if they pass through a common trace point, FUSE: o keeps its trace up to that point and
 *      takes w's from there (diffs = sum of the per-segment diffs); otherwise, if the paths never cross, drop the
 *      one whose box lies inside the other's box grown by BOX_FUZZ = 10.
 * Output order: (aread, abpos, bread, comp, order of survival) -- the reference's per-thread SORT_MAP order;
 * its cross-thread merge only compares (aread, abpos, thread slot), which coincides unless two records of
 * different threads tie on (aread, abpos) (SURVEY.md hard part 7).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"

#define TSPACE      100
#define BOX_FUZZ    10
#define ELIMINATED  0x4

typedef struct
  { int tlen, diffs, abpos, bbpos, aepos, bepos;
    unsigned flags;
    int aread, bread;
    uint8_t *trace;         /* points into the input byte pool or to owned memory */
    int owns;
    int64_t ord;            /* discovery order */
  } rec;

/* ---- how close two overlapping paths come, and whether they meet on a trace point ----------------------------------
 *
 * Restatement of what the reference's entwine (FastGA.c:2818-2941) computes, for `lo` starting at or before `hi` in A.
 * Both traces are sampled on the common TSPACE grid of A.  Between two consecutive grid lines each path advances by one
 * panel, so the signed B-separation (hi minus lo) is a running sum of per-panel deltas
 *
 *        sep(g+1) = sep(g) + ( hi.b[g] - lo.b[skip+g] ),        skip = panels of lo that lie before hi's first cell,
 *
 * framed by two interpolated samples: at hi's start (lo interpolated inside its current panel) and at the A coordinate
 * where the first of the two ends (the other interpolated -- always over a full TSPACE, which is what the reference's
 * arithmetic amounts to, its shorter-panel branch being unreachable).  The result is the separation of smallest
 * magnitude with its sign, 0 as soon as the paths touch or change sides; *meet is the last interior grid line on which
 * both paths pass through the same point (-1: none).
 */
static inline int nearer_zero(int best, int sep)
{ if (best > 0)
    return sep >= best ? best : (sep > 0 ? sep : 0);
  if (best < 0)
    return sep <= best ? best : (sep < 0 ? sep : 0);
  return 0;
}

static inline int panel_b(const rec *r, int panel)      /* B length of a trace panel */
{ return r->trace[2*panel+1]; }

static int path_gap(const rec *lo, const rec *hi, int *meet)
{ const int cell  = (hi->abpos/TSPACE)*TSPACE;                    /* grid line at or before hi's start           */
  const int skip  = hi->abpos/TSPACE - lo->abpos/TSPACE;          /* lo's panels that end at or before `cell`    */
  const int stop  = lo->aepos < hi->aepos ? lo->aepos : hi->aepos;
  const int lines = stop > cell ? (stop-cell-1)/TSPACE : 0;       /* grid lines strictly inside (cell, stop)      */
  int lob = lo->bbpos, hib = hi->bbpos;                           /* B coordinates of the two paths on the grid   */
  int best, g, last;

  for (g = 0; g < skip; g++)
    lob += panel_b(lo,g);
  /* first sample: at hi's start, lo interpolated within the panel it is in (its first panel may be a short one) */
  { const int span = skip == 0 ? cell+TSPACE - lo->abpos : TSPACE;
    const int from = skip == 0 ? lo->abpos : cell;
    best = hib - (lob + (panel_b(lo,skip) * (hi->abpos - from)) / span);
  }
  *meet = -1;
  for (g = 0; g < lines; g++)
    { const int sep = (hib += panel_b(hi,g)) - (lob += panel_b(lo,skip+g));
      best = nearer_zero(best,sep);
      if (sep == 0)
        *meet = cell + (g+1)*TSPACE;
    }
  /* last sample: where the first path ends; the other one is taken part of the way through its current panel */
  last = cell + lines*TSPACE;
  if (stop == lo->aepos)
    { hib += (panel_b(hi,lines) * (stop-last)) / TSPACE;
      lob  = lo->bepos;
    }
  else
    { lob += (panel_b(lo,skip+lines) * (stop-last)) / TSPACE;
      hib  = hi->bepos;
    }
  return nearer_zero(best,hib-lob);
}

static int by_abpos(const void *l, const void *r)
{ const rec *a = *(rec * const *) l, *b = *(rec * const *) r;
  if (a->abpos != b->abpos) return a->abpos - b->abpos;
  return (a->ord < b->ord) ? -1 : (a->ord > b->ord);
}

static int by_final(const void *l, const void *r)
{ const rec *a = *(rec * const *) l, *b = *(rec * const *) r;
  if (a->aread != b->aread) return a->aread - b->aread;
  if (a->abpos != b->abpos) return a->abpos - b->abpos;
  if (a->bread != b->bread) return a->bread - b->bread;
  if ((a->flags & 1) != (b->flags & 1)) return (int) (a->flags & 1) - (int) (b->flags & 1);
  return (a->ord < b->ord) ? -1 : (a->ord > b->ord);
}

static int by_discovery(const void *l, const void *r)
{ const fga_aln *a = l, *b = r;
  if (a->unit != b->unit) return a->unit < b->unit ? -1 : 1;
  return a->seq - b->seq;
}

/* ---- the elimination rules (FastGA.c:3440-3585) as predicates over a pair (lo, hi), lo.abpos <= hi.abpos ---------- */
enum { KEEP_BOTH = 0, DROP_HI, DROP_LO };

/* rule set 1: two records that share an end point.  Same start: the one reaching further in A stays (same box: the
 * reference compares lo's diffs with hi's aepos -- FastGA.c:3456 -- kept as it is); same end: the earlier start stays. */
static int shared_endpoint_rule(const rec *lo, const rec *hi)
{ const int same_start = lo->abpos == hi->abpos && lo->bbpos == hi->bbpos;
  const int same_end   = lo->aepos == hi->aepos && lo->bepos == hi->bepos;
  if (same_start)
    return ((same_end ? lo->diffs < hi->aepos : lo->aepos > hi->aepos)) ? DROP_HI : DROP_LO;
  if (same_end)
    return lo->abpos < hi->abpos ? DROP_HI : DROP_LO;
  return KEEP_BOTH;
}

/* rule set 2, for paths that stay apart: a record whose box lies inside the other's box grown by BOX_FUZZ goes; the
 * candidate is hi unless it is the longer of the two by more than the fuzz */
static int inside(const rec *in, const rec *box, int test_start)
{ return in->aepos <= box->aepos+BOX_FUZZ && in->bbpos >= box->bbpos-BOX_FUZZ && in->bepos <= box->bepos+BOX_FUZZ &&
         (!test_start || in->abpos >= box->abpos-BOX_FUZZ);
}

static int containment_rule(const rec *lo, const rec *hi)
{ if ((lo->aepos - lo->abpos) + BOX_FUZZ >= hi->aepos - hi->abpos)
    return inside(hi,lo,0) ? DROP_HI : KEEP_BOTH;       /* hi starts at or after lo: its start needs no test */
  return inside(lo,hi,1) ? DROP_LO : KEEP_BOTH;
}

/* lo and hi pass through the trace point at A = meet: lo continues along hi from there and hi goes */
static int fuse_at(rec *lo, rec *hi, int meet)
{ const int keep = 2 * ((meet - lo->abpos + TSPACE-1)/TSPACE);      /* bytes of lo's trace up to the meeting point  */
  const int from = 2 * ((meet - hi->abpos + TSPACE-1)/TSPACE);      /* first byte of hi's trace after it             */
  const int tlen = keep + (hi->tlen - from);
  uint8_t *t = malloc(tlen > 0 ? tlen : 1);
  int g, diffs = 0;
  if (t == NULL)
    return 1;
  memcpy(t,lo->trace,keep);
  memcpy(t+keep,hi->trace+from,hi->tlen-from);
  for (g = 0; g < tlen; g += 2)
    diffs += t[g];
  if (lo->owns) free(lo->trace);
  if (hi->owns) { free(hi->trace); hi->owns = 0; hi->trace = NULL; }
  lo->trace = t; lo->owns = 1;
  lo->tlen = tlen; lo->diffs = diffs;
  lo->aepos = hi->aepos; lo->bepos = hi->bepos;
  hi->flags |= ELIMINATED;
  return 0;
}

/* One contig pair and strand, seg[0..n) in abpos order.  Both sweeps take the records from the last to the first and
 * pair each with its live successors while their A intervals overlap; dropping the earlier record of a pair ends its
 * turn in the first sweep only (the reference's loops do the same). */
static int filter_segment(rec **seg, int n)
{ int sweep, at, nx;
  for (sweep = 0; sweep < 2; sweep++)
    for (at = n-1; at >= 0; at--)
      { rec *lo = seg[at];
        if (sweep == 1 && (lo->flags & ELIMINATED))
          continue;
        for (nx = at+1; nx < n && seg[nx]->abpos < lo->aepos; nx++)
          { rec *hi = seg[nx];
            int verdict, meet;
            if (hi->flags & ELIMINATED)
              continue;
            if (sweep == 0)
              verdict = shared_endpoint_rule(lo,hi);
            else
              { if (lo->bepos <= hi->bbpos || lo->bbpos >= hi->bepos)        /* disjoint in B */
                  continue;
                verdict = KEEP_BOTH;
                if (path_gap(lo,hi,&meet) != 0 && meet < 0)
                  verdict = containment_rule(lo,hi);
                if (meet >= 0 && fuse_at(lo,hi,meet))
                  return 1;
              }
            if (verdict == DROP_HI)
              hi->flags |= ELIMINATED;
            else if (verdict == DROP_LO)
              { lo->flags |= ELIMINATED;
                if (sweep == 0)
                  break;
              }
          }
      }
  return 0;
}

/* ---- helpers for large sets: the three orderings are O(n) / run on all threads ---- */

/* stable LSD radix sort of (key, value) pairs on the low `bits` bits of the key, 11 bits per pass */
static int sort_pairs(uint64_t *key, int64_t *val, int64_t n, int bits)
{ uint64_t *k2 = malloc(sizeof(uint64_t)*(n > 0 ? n : 1));
  int64_t  *v2 = malloc(sizeof(int64_t)*(n > 0 ? n : 1));
  int64_t  *cnt = malloc(sizeof(int64_t)*2048);
  uint64_t *ka = key, *kb = k2;
  int64_t  *va = val, *vb = v2;
  int shift;
  if (k2 == NULL || v2 == NULL || cnt == NULL)
    { free(k2); free(v2); free(cnt);
      return 1;
    }
  for (shift = 0; shift < bits; shift += 11)
    { int64_t i, sum = 0;
      memset(cnt,0,sizeof(int64_t)*2048);
      for (i = 0; i < n; i++)
        cnt[(ka[i] >> shift) & 0x7ff] += 1;
      if (n > 0 && cnt[(ka[0] >> shift) & 0x7ff] == n)
        continue;                                   /* this digit is the same everywhere */
      for (i = 0; i < 2048; i++)
        { int64_t c = cnt[i]; cnt[i] = sum; sum += c; }
      for (i = 0; i < n; i++)
        { int64_t d = cnt[(ka[i] >> shift) & 0x7ff]++;
          kb[d] = ka[i]; vb[d] = va[i];
        }
      { uint64_t *t = ka; ka = kb; kb = t; }
      { int64_t *t = va; va = vb; vb = t; }
    }
  if (ka != key)
    { memcpy(key,ka,sizeof(uint64_t)*n); memcpy(val,va,sizeof(int64_t)*n); }
  free(k2); free(v2); free(cnt);
  return 0;
}

typedef struct
  { rec    **perm;
    int64_t *segbeg;          /* nseg+1 */
    int64_t  nseg;
    int64_t *next;            /* shared cursor */
    pthread_mutex_t *lock;
    int      status;
  } seg_job;

static void *seg_thread(void *arg)
{ seg_job *J = arg;
  for (;;)
    { int64_t g, hi;
      pthread_mutex_lock(J->lock);
      g = *J->next;
      hi = g + 16 < J->nseg ? g + 16 : J->nseg;       /* a batch of segments per grab */
      *J->next = hi;
      pthread_mutex_unlock(J->lock);
      if (g >= J->nseg)
        break;
      for (; g < hi; g++)
        { const int64_t b = J->segbeg[g], e = J->segbeg[g+1];
          qsort(J->perm+b,e-b,sizeof(rec *),by_abpos);
          if (filter_segment(J->perm+b,(int) (e-b)))
            J->status = 1;
        }
    }
  return NULL;
}

static int run_segments(rec **perm, int64_t *segbeg, int64_t nseg, int nthreads)
{ pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
  seg_job   job[64];
  pthread_t th[64];
  int64_t   next = 0;
  int t, rc = 0;
  if (nthreads > 64) nthreads = 64;
  if (nthreads < 1 || nseg < 64 || segbeg[nseg] < 50000) nthreads = 1;     /* starting threads costs ~1 ms */
  for (t = 0; t < nthreads; t++)
    { job[t].perm = perm; job[t].segbeg = segbeg; job[t].nseg = nseg; job[t].next = &next; job[t].lock = &lock;
      job[t].status = 0;
    }
  for (t = 1; t < nthreads; t++)
    if (pthread_create(th+t,NULL,seg_thread,job+t) != 0)
      { seg_thread(job+t); th[t] = 0; }
  seg_thread(job);
  for (t = 1; t < nthreads; t++)
    if (th[t]) pthread_join(th[t],NULL);
  for (t = 0; t < nthreads; t++)
    rc |= job[t].status;
  return rc;
}

/* in: alignments as produced by fga_extend (any order); out: filtered + finally ordered copy */
int fga_filter_alignments(const fga_alns *in, fga_alns **out)
{ return fga_filter_alignments_mt(in,1,out); }

int fga_filter_alignments_mt(const fga_alns *in, int nthreads, fga_alns **out)
{ fga_alns *R;
  fga_aln  *sorted = NULL;
  rec      *recs = NULL, **perm = NULL, **live = NULL;
  uint64_t *skey = NULL;
  int64_t  *sval = NULL, *segbeg = NULL;
  int64_t   n = in->naln, i, j, nlive = 0, tbytes = 0, off, nseg = 0;

  *out = NULL;
  R = calloc(1,sizeof(fga_alns));
  if (R == NULL) goto oom;
  R->ncalls = in->ncalls; R->nwaves = in->nwaves;
  if (n == 0)
    { R->alns = malloc(sizeof(fga_aln)); R->tbytes = malloc(16);
      *out = R;
      return 0;
    }
  sorted = malloc(sizeof(fga_aln)*n);
  recs   = malloc(sizeof(rec)*n);
  perm   = malloc(sizeof(rec *)*n);
  live   = malloc(sizeof(rec *)*n);
  skey   = malloc(sizeof(uint64_t)*n);
  sval   = malloc(sizeof(int64_t)*n);
  segbeg = malloc(sizeof(int64_t)*(n+1));
  if (sorted == NULL || recs == NULL || perm == NULL || live == NULL || skey == NULL || sval == NULL || segbeg == NULL)
    goto oom;

  /* discovery order = (unit, seq): radix sort when the two fit a 64-bit key, else the comparison sort */
  { int64_t maxu = 0, maxs = 0;
    int ub = 1, sb = 1, ok = 1;
    for (i = 0; i < n; i++)
      { if (in->alns[i].unit < 0 || in->alns[i].seq < 0) { ok = 0; break; }
        if (in->alns[i].unit > maxu) maxu = in->alns[i].unit;
        if (in->alns[i].seq > maxs) maxs = in->alns[i].seq;
      }
    while (((int64_t) 1 << ub) <= maxu) ub += 1;
    while (((int64_t) 1 << sb) <= maxs) sb += 1;
    if (ok)
      { for (i = 0; i < n; i++)
          { skey[i] = ((uint64_t) in->alns[i].unit << sb) | (uint64_t) in->alns[i].seq;
            sval[i] = i;
          }
        if (sort_pairs(skey,sval,n,ub+sb)) goto oom;
        for (i = 0; i < n; i++)
          sorted[i] = in->alns[sval[i]];
      }
    else
      { memcpy(sorted,in->alns,sizeof(fga_aln)*n);
        qsort(sorted,n,sizeof(fga_aln),by_discovery);
      }
  }
  for (i = 0; i < n; i++)
    { rec *r = recs+i;
      const fga_aln *a = sorted+i;
      r->tlen = a->tlen; r->diffs = a->diffs; r->abpos = a->abpos; r->bbpos = a->bbpos;
      r->aepos = a->aepos; r->bepos = a->bepos; r->flags = a->flags; r->aread = a->aread; r->bread = a->bread;
      r->trace = in->tbytes + a->toff; r->owns = 0; r->ord = i;
      perm[i] = r;
    }
  /* segments = runs of equal (aread, bread, comp) in discovery order (units are key-ordered); independent of each other */
  for (i = 0; i < n; i = j)
    { for (j = i+1; j < n; j++)
        if (recs[j].aread != recs[i].aread || recs[j].bread != recs[i].bread ||
            (recs[j].flags & 1) != (recs[i].flags & 1))
          break;
      segbeg[nseg++] = i;
    }
  segbeg[nseg] = n;
  if (run_segments(perm,segbeg,nseg,nthreads)) goto oom;
  for (i = 0; i < n; i++)
    if (!(perm[i]->flags & ELIMINATED))
      { perm[i]->ord = nlive;                 /* order of survival = the reference's file order */
        live[nlive++] = perm[i];
        tbytes += perm[i]->tlen;
      }

  /* final order (aread, abpos, bread, comp, survival): radix sort on (aread, abpos), then the few runs that tie on
   * both are put in order by the rest of the key */
  if (nlive > 1)
    { for (i = 0; i < nlive; i++)
        { skey[i] = ((uint64_t) (uint32_t) live[i]->aread << 32) | (uint32_t) live[i]->abpos; sval[i] = i; }
      if (sort_pairs(skey,sval,nlive,64)) goto oom;
      for (i = 0; i < nlive; i++)
        perm[i] = live[sval[i]];
      for (i = 0; i < nlive; i = j)
        { for (j = i+1; j < nlive && skey[j] == skey[i]; j++)
            ;
          if (j-i > 1)
            qsort(perm+i,j-i,sizeof(rec *),by_final);
        }
      memcpy(live,perm,sizeof(rec *)*nlive);
    }

  R->naln = nlive; R->ntrace = tbytes;
  R->alns = malloc(sizeof(fga_aln)*(nlive+1));
  R->tbytes = malloc(tbytes+16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;
  off = 0;
  for (i = 0; i < nlive; i++)
    { rec *r = live[i];
      fga_aln *a = R->alns+i;
      memset(a,0,sizeof(*a));
      a->tlen = r->tlen; a->diffs = r->diffs; a->abpos = r->abpos; a->bbpos = r->bbpos;
      a->aepos = r->aepos; a->bepos = r->bepos; a->flags = r->flags & 0x3; a->aread = r->aread; a->bread = r->bread;
      a->unit = -1; a->seq = (int32_t) i; a->toff = off;
      memcpy(R->tbytes+off,r->trace,r->tlen);
      off += r->tlen;
    }
  for (i = 0; i < n; i++)
    if (recs[i].owns) free(recs[i].trace);
  free(sorted); free(recs); free(perm); free(live); free(skey); free(sval); free(segbeg);
  *out = R;
  return 0;

oom:
  fga_set_error("out of memory in alignment filter");
  free(sorted); free(recs); free(perm); free(live); free(skey); free(sval); free(segbeg);
  if (R != NULL) { free(R->alns); free(R->tbytes); free(R); }
  return 1;
}
