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

/* the O(n) passes of the driver, one slice per thread */
typedef struct
  { const fga_alns *in;
    const fga_aln *sorted;
    rec       *recs, **perm, **live;
    uint64_t  *skey;
    int64_t   *sval;
    int        sb;                       /* bits of the seq field in the discovery key */
    fga_alns  *R;
    int64_t   *off;                      /* trace offset of every surviving record */
    int64_t   *segbeg, *segout;          /* segments of perm; first survivor slot of every segment (nseg+1) */
    int32_t   *abp;                      /* abpos of the survivors, beside live */
    int64_t    tsum[65];                 /* trace bytes of the slices before a thread's slice of the survivors */
  } pass_ctx;

static void pass_discovery_keys(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t i;
  (void) id;
  for (i = b; i < e; i++)
    { C->skey[i] = ((uint64_t) C->in->alns[i].unit << C->sb) | (uint64_t) C->in->alns[i].seq;
      C->sval[i] = i;
    }
}

static void pass_records(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t i;
  (void) id;
  for (i = b; i < e; i++)
    { rec *r = C->recs+i;
      const fga_aln *a = C->sorted != NULL ? C->sorted+i : C->in->alns + C->sval[i];
      r->tlen = a->tlen; r->diffs = a->diffs; r->abpos = a->abpos; r->bbpos = a->bbpos;
      r->aepos = a->aepos; r->bepos = a->bepos; r->flags = a->flags; r->aread = a->aread; r->bread = a->bread;
      r->trace = C->in->tbytes + a->toff; r->owns = 0; r->ord = i;
      C->perm[i] = r;
    }
}

/* survivors of every segment: counted, then laid out segment by segment (slices of SEGMENTS) */
static void pass_count_live(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t s, i;
  (void) id;
  for (s = b; s < e; s++)
    { int64_t c = 0;
      for (i = C->segbeg[s]; i < C->segbeg[s+1]; i++)
        c += !(C->perm[i]->flags & ELIMINATED);
      C->segout[s] = c;
    }
}

static void pass_fill_live(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t s, i;
  (void) id;
  for (s = b; s < e; s++)
    { int64_t o = C->segout[s];
      for (i = C->segbeg[s]; i < C->segbeg[s+1]; i++)
        if (!(C->perm[i]->flags & ELIMINATED))
          { C->perm[i]->ord = o;                 /* order of survival = the reference's file order */
            C->abp[o] = C->perm[i]->abpos;
            C->live[o++] = C->perm[i];
          }
    }
}

static void pass_trace_sums(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t i, t = 0;
  for (i = b; i < e; i++)
    t += C->live[i]->tlen;
  C->tsum[id+1] = t;
}

static void pass_copy_out(void *arg, int id, int64_t b, int64_t e)
{ pass_ctx *C = arg;
  int64_t i, off = C->tsum[id];
  for (i = b; i < e; i++)
    { rec *r = C->live[i];
      fga_aln *a = C->R->alns+i;
      memset(a,0,sizeof(*a));
      a->tlen = r->tlen; a->diffs = r->diffs; a->abpos = r->abpos; a->bbpos = r->bbpos;
      a->aepos = r->aepos; a->bepos = r->bepos; a->flags = r->flags & 0x3; a->aread = r->aread; a->bread = r->bread;
      a->unit = -1; a->seq = (int32_t) i; a->toff = off;
      memcpy(C->R->tbytes + off,r->trace,r->tlen);
      off += r->tlen;
    }
}


/* ---- final order by merging --------------------------------------------------------------------------------------
 * The survivors of a segment -- one (aread, bread, comp) -- leave the filter in (abpos, survival) order, so the final
 * order (aread, abpos, bread, comp, survival) of an A contig is the MERGE of its segments' lists by (abpos, bread, comp):
 * O(n log k) comparisons on 8-byte pointers instead of two radix sorts over 16-byte pairs with their key builds and
 * gathers (at 2 M records of a 3 Gbp part: 2 x 70 ms of the 260 the last part's filter adds to the run).  A contig's
 * merge is cut by abpos splitters (sampled from its lists) into tasks of similar size; threads take tasks from a counter. */
typedef struct
  { int64_t beg, end;        /* the segment's survivors: live[beg,end) */
    int32_t aread, key2;     /* key2 = bread << 1 | comp */
  } seg_list;

typedef struct
  { int64_t l0, l1;          /* lists [l0,l1) of `lists` (one A contig) */
    int32_t lo, hi;          /* abpos range [lo,hi) */
    int64_t out;             /* first output position */
  } merge_task;

typedef struct
  { rec        **live, **out;
    const int32_t *abp;      /* abpos of live[i] */
    seg_list    *lists;
    merge_task  *task;
    int64_t      ntask, next;
    int          failed;
  } merge_ctx;

static int by_aread_key2(const void *l, const void *r)
{ const seg_list *a = l, *b = r;
  if (a->aread != b->aread) return a->aread < b->aread ? -1 : 1;
  if (a->key2 != b->key2) return a->key2 < b->key2 ? -1 : 1;
  return a->beg < b->beg ? -1 : (a->beg > b->beg);
}

static int by_int32(const void *l, const void *r)
{ const int32_t a = *(const int32_t *) l, b = *(const int32_t *) r;
  return a < b ? -1 : (a > b);
}

static inline int64_t first_at_or_after(const int32_t *abp, int64_t b, int64_t e, int32_t x)     /* first i in [b,e) with abpos >= x */
{ while (b < e)
    { const int64_t m = b + ((e-b) >> 1);
      if (abp[m] < x) b = m+1; else e = m;
    }
  return b;
}

typedef struct { int32_t abpos, key2; int64_t at, end; } heap_item;       /* `at` also orders lists of one key (there is one) */

static inline int heap_less(const heap_item *a, const heap_item *b)
{ if (a->abpos != b->abpos) return a->abpos < b->abpos;
  if (a->key2 != b->key2) return a->key2 < b->key2;
  return a->at < b->at;
}

static void heap_down(heap_item *h, int n, int i)
{ const heap_item x = h[i];
  for (;;)
    { int c = 2*i+1;
      if (c >= n) break;
      if (c+1 < n && heap_less(h+c+1,h+c)) c += 1;
      if (!heap_less(h+c,&x)) break;
      h[i] = h[c];
      i = c;
    }
  h[i] = x;
}

static void merge_tasks(void *arg, int id, int64_t b0, int64_t e0)
{ merge_ctx *M = arg;
  heap_item *h = NULL;
  int64_t hcap = 0;
  (void) id; (void) b0; (void) e0;
  for (;;)
    { const int64_t t = __sync_fetch_and_add(&M->next,1);
      const merge_task *T;
      int64_t l, o;
      int n = 0, i;
      if (t >= M->ntask)
        break;
      T = M->task+t;
      if (T->l1-T->l0 > hcap)
        { hcap = 2*(T->l1-T->l0) + 64;
          free(h);
          h = malloc(sizeof(heap_item)*hcap);
          if (h == NULL) { M->failed = 1; break; }
        }
      for (l = T->l0; l < T->l1; l++)
        { const seg_list *S = M->lists+l;
          const int64_t b = first_at_or_after(M->abp,S->beg,S->end,T->lo);
          const int64_t e = T->hi == 0x7fffffff ? S->end : first_at_or_after(M->abp,b,S->end,T->hi);
          if (b < e)
            { h[n].abpos = M->abp[b]; h[n].key2 = S->key2; h[n].at = b; h[n].end = e; n += 1; }
        }
      o = T->out;
      if (n == 1)
        { memcpy(M->out+o,M->live+h[0].at,sizeof(rec *)*(h[0].end-h[0].at));
          continue;
        }
      for (i = n/2-1; i >= 0; i--)
        heap_down(h,n,i);
      while (n > 0)
        { M->out[o++] = M->live[h[0].at];
          if (++h[0].at < h[0].end)
            h[0].abpos = M->abp[h[0].at];
          else
            h[0] = h[--n];
          if (n > 1)
            heap_down(h,n,0);
        }
    }
  free(h);
}

/* live[0,nlive) segment by segment (lists[0,nlist): every list in (abpos, survival) order) -> out in the final order.
 * Returns 0, or 1 when out of memory. */
static int final_order_by_merge(fga_team *team, rec **live, const int32_t *abp, rec **out, int64_t nlive, seg_list *lists,
                                int64_t nlist)
{ merge_ctx   M;
  merge_task *task = NULL;
  int32_t    *sample = NULL;
  int64_t     ntask = 0, tcap, g0, g1, base = 0, chunk, l;
  const int   nthr = fga_team_size(team);

  qsort(lists,nlist,sizeof(seg_list),by_aread_key2);
  chunk = nlive/(8*(int64_t) nthr);
  if (chunk < 8192) chunk = 8192;
  { const char *ev = getenv("FGA_FILTER_CHUNK");       /* test hook: records per merge task */
    if (ev != NULL && atoll(ev) > 0) chunk = atoll(ev);
  }
  tcap = 2*nlist + nlive/chunk + 16;
  task = malloc(sizeof(merge_task)*tcap);
  if (task == NULL) return 1;
  for (g0 = 0; g0 < nlist; g0 = g1)
    { int64_t m = 0, parts, k;
      for (g1 = g0; g1 < nlist && lists[g1].aread == lists[g0].aread; g1++)
        m += lists[g1].end-lists[g1].beg;
      parts = (m + chunk - 1)/chunk;
      if (parts <= 1)
        { task[ntask].l0 = g0; task[ntask].l1 = g1; task[ntask].lo = -0x7fffffff-1; task[ntask].hi = 0x7fffffff;
          task[ntask].out = base; ntask += 1;
        }
      else
        { /* splitters: every step-th abpos of every list, sorted, at equal ranks */
          const int64_t step = m/(parts*64) > 0 ? m/(parts*64) : 1;
          int64_t ns = 0, scap = m/step + (g1-g0) + 16;
          int32_t lo = -0x7fffffff-1;
          int32_t *sm = realloc(sample,sizeof(int32_t)*scap);
          if (sm == NULL) { free(sample); free(task); return 1; }
          sample = sm;
          for (l = g0; l < g1; l++)
            for (k = lists[l].beg + step/2; k < lists[l].end; k += step)
              sample[ns++] = abp[k];
          qsort(sample,ns,sizeof(int32_t),by_int32);
          for (k = 1; k <= parts; k++)
            { const int32_t hi = (k == parts || ns == 0) ? 0x7fffffff : sample[(ns*k)/parts < ns ? (ns*k)/parts : ns-1];
              int64_t off = 0;
              if (hi <= lo && k < parts)
                continue;                                   /* equal splitters: the range is empty */
              for (l = g0; l < g1; l++)
                off += first_at_or_after(abp,lists[l].beg,lists[l].end,lo) - lists[l].beg;
              task[ntask].l0 = g0; task[ntask].l1 = g1; task[ntask].lo = lo; task[ntask].hi = hi;
              task[ntask].out = base + off; ntask += 1;
              lo = hi;
              if (hi == 0x7fffffff)
                break;
            }
        }
      base += m;
    }
  M.live = live; M.abp = abp; M.out = out; M.lists = lists; M.task = task; M.ntask = ntask; M.next = 0; M.failed = 0;
  fga_team_run(team,nthr,merge_tasks,&M);
  free(task); free(sample);
  return M.failed;
}

int fga_filter_alignments_mt(const fga_alns *in, int nthreads, fga_alns **out)
{ fga_alns *R;
  fga_aln  *sorted = NULL;
  rec      *recs = NULL, **perm = NULL, **live = NULL;
  uint64_t *skey = NULL;
  int64_t  *sval = NULL, *segbeg = NULL;
  seg_list *lists = NULL;
  int64_t  *segout = NULL;
  int32_t  *abp = NULL;
  int64_t   n = in->naln, i, j, nlive = 0, tbytes = 0, nseg = 0, nlist = 0;
  fga_team *team = NULL;
  pass_ctx  C;

  const int timing = getenv("FGA_FILTER_TIMING") != NULL;
  double tw[6] = {0,0,0,0,0,0};
  *out = NULL;
  R = calloc(1,sizeof(fga_alns));
  if (R == NULL) goto oom;
  R->ncalls = in->ncalls; R->nwaves = in->nwaves;
  tw[0] = fga_wall();
  if (n == 0)
    { R->alns = malloc(sizeof(fga_aln)); R->tbytes = malloc(16);
      *out = R;
      return 0;
    }
  team   = fga_team_open(n < 50000 ? 1 : nthreads);         /* starting threads costs ~1 ms */
  recs   = malloc(sizeof(rec)*n);
  perm   = malloc(sizeof(rec *)*n);
  live   = malloc(sizeof(rec *)*n);
  skey   = malloc(sizeof(uint64_t)*n);
  sval   = malloc(sizeof(int64_t)*(n+1));
  segbeg = malloc(sizeof(int64_t)*(n+1));
  if (team == NULL || recs == NULL || perm == NULL || live == NULL || skey == NULL || sval == NULL || segbeg == NULL)
    goto oom;
  memset(&C,0,sizeof(C));
  C.in = in; C.recs = recs; C.perm = perm; C.live = live; C.skey = skey; C.sval = sval; C.R = R;

  /* discovery order = (unit, seq): radix sort when the two fit a 64-bit key, else the comparison sort */
  { int64_t maxu = 0, maxs = 0;
    int ub = 1, sb = 1, ok = 1, inorder = 1;
    for (i = 0; i < n; i++)
      { if (in->alns[i].unit < 0 || in->alns[i].seq < 0) { ok = 0; break; }
        if (in->alns[i].unit > maxu) maxu = in->alns[i].unit;
        if (in->alns[i].seq > maxs) maxs = in->alns[i].seq;
        if (i > 0 && (in->alns[i].unit < in->alns[i-1].unit ||
                      (in->alns[i].unit == in->alns[i-1].unit && in->alns[i].seq < in->alns[i-1].seq)))
          inorder = 0;
      }
    while (((int64_t) 1 << ub) <= maxu) ub += 1;
    while (((int64_t) 1 << sb) <= maxs) sb += 1;
    if (ok && inorder)                      /* fga_extend hands its records over in discovery order (sorted on the device) */
      C.sorted = in->alns;
    else if (ok && ub + sb <= 64)
      { C.sb = sb;
        fga_team_run(team,n,pass_discovery_keys,&C);
        if (fga_team_sort_pairs(team,skey,sval,n,ub+sb)) goto oom;
      }
    else
      { sorted = malloc(sizeof(fga_aln)*n);
        if (sorted == NULL) goto oom;
        memcpy(sorted,in->alns,sizeof(fga_aln)*n);
        qsort(sorted,n,sizeof(fga_aln),by_discovery);
        C.sorted = sorted;
      }
  }
  tw[1] = fga_wall();
  fga_team_run(team,n,pass_records,&C);
  /* segments = runs of equal (aread, bread, comp) in discovery order (units are key-ordered); independent of each other */
  for (i = 0; i < n; i = j)
    { for (j = i+1; j < n; j++)
        if (recs[j].aread != recs[i].aread || recs[j].bread != recs[i].bread ||
            (recs[j].flags & 1) != (recs[i].flags & 1))
          break;
      segbeg[nseg++] = i;
    }
  segbeg[nseg] = n;
  tw[2] = fga_wall();
  if (run_segments(perm,segbeg,nseg,fga_team_size(team) > 1 ? nthreads : 1)) goto oom;
  tw[3] = fga_wall();
  lists  = malloc(sizeof(seg_list)*(nseg > 0 ? nseg : 1));
  segout = malloc(sizeof(int64_t)*(nseg+1));
  abp    = malloc(sizeof(int32_t)*(n > 0 ? n : 1));
  if (lists == NULL || segout == NULL || abp == NULL) goto oom;
  C.segbeg = segbeg; C.segout = segout; C.abp = abp;
  fga_team_run(team,nseg,pass_count_live,&C);
  for (j = 0; j < nseg; j++)
    { const int64_t c = segout[j];
      segout[j] = nlive;
      if (c > 0)
        { lists[nlist].beg = nlive; lists[nlist].end = nlive+c;
          nlist += 1;
        }
      nlive += c;
    }
  segout[nseg] = nlive;
  fga_team_run(team,nseg,pass_fill_live,&C);
  for (j = 0; j < nlist; j++)
    { const rec *r = live[lists[j].beg];
      lists[j].aread = r->aread;
      lists[j].key2 = (int32_t) (((uint32_t) r->bread << 1) | (r->flags & 1));
    }

  /* final order (aread, abpos, bread, comp, survival): the merge of every A contig's segment lists */
  if (nlive > 1)
    { if (final_order_by_merge(team,live,abp,perm,nlive,lists,nlist)) goto oom;
      memcpy(live,perm,sizeof(rec *)*nlive);
    }

  tw[4] = fga_wall();
  C.tsum[0] = 0;
  fga_team_run(team,nlive,pass_trace_sums,&C);
  for (j = 0; j < fga_team_size(team); j++)
    C.tsum[j+1] += C.tsum[j];
  tbytes = C.tsum[fga_team_size(team)];
  R->naln = nlive; R->ntrace = tbytes;
  R->alns = malloc(sizeof(fga_aln)*(nlive+1));
  R->tbytes = malloc(tbytes+16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;
  fga_team_run(team,nlive,pass_copy_out,&C);
  for (i = 0; i < n; i++)
    if (recs[i].owns) free(recs[i].trace);
  free(sorted); free(recs); free(perm); free(live); free(skey); free(sval); free(segbeg); free(lists); free(segout); free(abp);
  if (timing)
    fprintf(stderr,"filter timing: %lld records, %lld segments: discovery order %.1f ms, records %.1f ms, segments %.1f ms (%d threads), "
                   "final order %.1f ms, copy out %.1f ms\n",(long long) n,(long long) nseg,1e3*(tw[1]-tw[0]),1e3*(tw[2]-tw[1]),
            1e3*(tw[3]-tw[2]),fga_team_size(team),1e3*(tw[4]-tw[3]),1e3*(fga_wall()-tw[4]));
  fga_team_close(team);
  *out = R;
  return 0;

oom:
  fga_set_error("out of memory in alignment filter");
  free(sorted); free(recs); free(perm); free(live); free(skey); free(sval); free(segbeg); free(lists); free(segout); free(abp);
  fga_team_close(team);
  if (R != NULL) { free(R->alns); free(R->tbytes); free(R); }
  return 1;
}

/* ---- the filtered sets of several parts as one set in final order ------------------------------------------------------
 * Each input is an output of fga_filter_alignments[_mt]: in final order (aread, abpos, bread, comp, survival).  A-contig
 * parts are disjoint in aread, so the final order of the union is the inputs' runs of equal aread laid out by aread --
 * copies, no sort (what la_merge does with the per-thread files of the reference, FastGA.c:3991-4133).  Inputs that do
 * share an A contig are merged run against run on the rest of the key, the earlier input first on ties. */
typedef struct { int32_t aread; int set; int64_t beg, cnt; } arun;

static int arun_cmp(const void *l, const void *r)
{ const arun *a = l, *b = r;
  if (a->aread != b->aread) return a->aread < b->aread ? -1 : 1;
  return a->set - b->set;
}

static int aln_minor_cmp(const fga_aln *a, const fga_aln *b)
{ if (a->abpos != b->abpos) return a->abpos < b->abpos ? -1 : 1;
  if (a->bread != b->bread) return a->bread < b->bread ? -1 : 1;
  if ((a->flags & 1) != (b->flags & 1)) return (a->flags & 1) < (b->flags & 1) ? -1 : 1;
  return 0;
}

/* a piece of a run that comes from one input: records [x0,x1) of the run to position `at`, trace bytes to `tat` */
typedef struct { const fga_alns *S; int64_t beg, x0, x1, at, tat; } copy_task;
typedef struct { copy_task *task; int64_t ntask, next; fga_alns *R; } copy_ctx;

static void copy_tasks(void *arg, int id, int64_t b0, int64_t e0)
{ copy_ctx *C = arg;
  (void) id; (void) b0; (void) e0;
  for (;;)
    { const int64_t t = __sync_fetch_and_add(&C->next,1);
      const copy_task *T;
      const fga_aln *src;
      int64_t t0, t1, x, at;
      if (t >= C->ntask)
        break;
      T = C->task+t;
      src = T->S->alns + T->beg;
      t0 = src[T->x0].toff; t1 = src[T->x1-1].toff + src[T->x1-1].tlen;
      memcpy(C->R->tbytes + T->tat,T->S->tbytes + t0,(size_t) (t1 - t0));   /* a filtered set's trace bytes are laid out in record order */
      at = T->at;
      for (x = T->x0; x < T->x1; x++)
        { fga_aln a = src[x];
          a.toff = T->tat + (a.toff - t0); a.seq = (int32_t) at; a.unit = -1;
          C->R->alns[at++] = a;
        }
    }
}

int fga_alns_merge_filtered(const fga_alns *const *fin, int nfin, fga_alns **out)
{ return fga_alns_merge_filtered_mt(fin,nfin,1,out); }

int fga_alns_merge_filtered_mt(const fga_alns *const *fin, int nfin, int nthreads, fga_alns **out)
{ fga_alns *R = calloc(1,sizeof(fga_alns));
  arun *runs = NULL;
  copy_task *task = NULL;
  int64_t nrun = 0, caprun = 0, at = 0, tat = 0, i, ntask = 0, captask = 0;
  int k;
  *out = NULL;
  if (R == NULL) goto oom;
  for (k = 0; k < nfin; k++)
    if (fin[k] != NULL)
      { R->naln += fin[k]->naln; R->ntrace += fin[k]->ntrace; R->ncalls += fin[k]->ncalls; R->nwaves += fin[k]->nwaves;
        for (i = 0; i < fin[k]->naln; )
          { int64_t j = i+1;
            while (j < fin[k]->naln && fin[k]->alns[j].aread == fin[k]->alns[i].aread) j += 1;
            if (nrun >= caprun)
              { arun *nr;
                caprun = caprun ? 2*caprun : 1024;
                nr = realloc(runs,sizeof(arun)*caprun);
                if (nr == NULL) goto oom;
                runs = nr;
              }
            runs[nrun].aread = fin[k]->alns[i].aread; runs[nrun].set = k; runs[nrun].beg = i; runs[nrun].cnt = j-i;
            nrun += 1;
            i = j;
          }
      }
  R->alns = malloc(sizeof(fga_aln)*(R->naln+1));
  R->tbytes = malloc(R->ntrace+16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;
  if (nrun > 1)
    qsort(runs,nrun,sizeof(arun),arun_cmp);
  for (i = 0; i < nrun; )
    { int64_t j = i+1;
      while (j < nrun && runs[j].aread == runs[i].aread) j += 1;
      if (j == i+1)                                   /* the usual case: the contig's records come from one part: copied */
        { const fga_alns *S = fin[runs[i].set];        /* in pieces of 64 k records by all threads, below             */
          const fga_aln *src = S->alns + runs[i].beg;
          int64_t x0;
          for (x0 = 0; x0 < runs[i].cnt; x0 += 65536)
            { const int64_t x1 = x0 + 65536 < runs[i].cnt ? x0 + 65536 : runs[i].cnt;
              if (ntask >= captask)
                { copy_task *nt;
                  captask = captask ? 2*captask : 1024;
                  nt = realloc(task,sizeof(copy_task)*captask);
                  if (nt == NULL) goto oom;
                  task = nt;
                }
              task[ntask].S = S; task[ntask].beg = runs[i].beg; task[ntask].x0 = x0; task[ntask].x1 = x1;
              task[ntask].at = at; task[ntask].tat = tat;
              ntask += 1;
              at += x1-x0;
              tat += (src[x1-1].toff + src[x1-1].tlen) - src[x0].toff;
            }
        }
      else                                            /* several inputs hold records of this contig: merge their runs */
        { int64_t *pos = calloc(j-i,sizeof(int64_t));
          if (pos == NULL) goto oom;
          for (;;)
            { int64_t best = -1, x;
              for (x = i; x < j; x++)
                if (pos[x-i] < runs[x].cnt &&
                    (best < 0 || aln_minor_cmp(fin[runs[x].set]->alns + runs[x].beg + pos[x-i],
                                               fin[runs[best].set]->alns + runs[best].beg + pos[best-i]) < 0))
                  best = x;
              if (best < 0) break;
              { const fga_alns *S = fin[runs[best].set];
                fga_aln a = S->alns[runs[best].beg + pos[best-i]];
                memcpy(R->tbytes + tat,S->tbytes + a.toff,(size_t) a.tlen);
                a.toff = tat; a.seq = (int32_t) at; a.unit = -1;
                R->alns[at++] = a;
                tat += a.tlen;
                pos[best-i] += 1;
              }
            }
          free(pos);
        }
      i = j;
    }
  if (ntask > 0)
    { copy_ctx C;
      fga_team *team = fga_team_open(at < 50000 ? 1 : nthreads);
      if (team == NULL) goto oom;
      C.task = task; C.ntask = ntask; C.next = 0; C.R = R;
      fga_team_run(team,fga_team_size(team),copy_tasks,&C);
      fga_team_close(team);
    }
  free(task); task = NULL;
  R->naln = at; R->ntrace = tat;
  free(runs);
  *out = R;
  return 0;
oom:
  fga_set_error("out of memory merging filtered alignment sets");
  free(runs); free(task);
  if (R != NULL) { free(R->alns); free(R->tbytes); free(R); }
  return 1;
}
