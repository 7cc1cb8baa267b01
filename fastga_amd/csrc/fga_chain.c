/* fga_chain.c -- diagonal-band chain detection over the sorted seed records (host, multi-threaded).
 *
 * Replaces the chain scan of align_contigs (reference FastGA.c:3016-3176, 3340-3403) for every
 * (strand, A contig, B contig) run of the sorted 128-bit keys produced by fga_seed_sort.
 * Semantics:
 *   records of one run are grouped by diagonal bucket d = diag>>6.  Each present bucket d forms a *unit*
 *   together with bucket d+1 when that one is present (aux); a unit whose bucket d was already the d+1 half
 *   of the previous unit (new = 0) and that has no d+1 partner is skipped.  Inside a unit the two buckets
 *   are merged by anti-diagonal (ties: bucket d first) and scanned once: a chain continues while
 *   anti < ahgh + CHAIN_BREAK with ahgh = max(anti + 2*lcp); coverage adds the not-yet-covered part of
 *   [anti, anti+2*lcp); when a chain ends with cov >= CHAIN_MIN and (mix != 1 || new) it is a *hit*
 *   (dgmin,dgmax,alow,ahgh), shifted to contig coordinates (FastGA.c:3205-3216).
 * Units are independent of one another (the `alast` state of the extension loop is per unit), which is what
 * the GPU extension stage parallelises over; hits of one unit must stay in order.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"

#define BUCK_SHIFT  6
#define BUCK_WIDTH  64

typedef struct { uint64_t lo, hi; } key128;

typedef struct
  { int wa, wb, wd, wt;
    int s_anti, s_buck, s_b, s_a, s_strand;     /* bit offsets */
  } layout;

static inline uint64_t field(const key128 *k, int shift, int width)
{ uint64_t v;
  if (shift >= 64)
    v = k->hi >> (shift-64);
  else
    { v = k->lo >> shift;
      if (shift > 0 && shift+width > 64)
        v |= k->hi << (64-shift);
    }
  if (width < 64)
    v &= (((uint64_t) 1) << width) - 1;
  return v;
}

/* segment id = everything above the bucket field (strand, A contig, B contig) */
static inline uint64_t segid(const key128 *k, const layout *L)
{ return field(k,L->s_b,L->wb + L->wa + 1); }

typedef struct
  { fga_hit  *hits;  int64_t nhit, mhit;
    fga_unit *units; int64_t nunit, munit;
  } outvec;

static int push_hit(outvec *o, const fga_hit *h)
{ if (o->nhit >= o->mhit)
    { o->mhit = o->mhit*2 + 1024;
      o->hits = realloc(o->hits,sizeof(fga_hit)*o->mhit);
      if (o->hits == NULL) return 1;
    }
  o->hits[o->nhit++] = *h;
  return 0;
}

static int push_unit(outvec *o, const fga_unit *u)
{ if (o->nunit >= o->munit)
    { o->munit = o->munit*2 + 256;
      o->units = realloc(o->units,sizeof(fga_unit)*o->munit);
      if (o->units == NULL) return 1;
    }
  o->units[o->nunit++] = *u;
  return 0;
}

typedef struct
  { const key128 *keys;
    int64_t       n, c0, c1;
    layout        L;
    const fga_chain_params *prm;
    outvec        out;
    int           status;
  } targ;

/* The chains of one unit: bucket run [b,m) (diag>>6 = d, side 1) and, if any, run [m,e) (d+1, side 2).
 *
 * Closed form of the reference's scan (FastGA.c:3086-3176, 3353-3368; the same form fga_chain.hip evaluates with wave
 * scans).  Take the records of both runs in anti-diagonal order, run d first on ties.  With
 *     top(i)   = anti(i) + 2 lcp(i)                 reach(i) = max top(j), j < i        (-CHAIN_BREAK before the first)
 * record i OPENS a chain iff anti(i) >= reach(i) + CHAIN_BREAK (reach is a plain prefix maximum: a chain's first top
 * already exceeds everything before it), otherwise it extends the open one and adds max(0, top(i) - max(reach(i),anti(i)))
 * to its coverage.  A chain is reported when the next one opens or the unit ends, if its coverage reaches CHAIN_MIN
 * and it is not a pure run-d chain already seen as the d+1 half of the previous unit (sides == 1 && !isnew).
 */
typedef struct
  { int64_t alow, reach, cov;
    int     dgmin, dgmax, sides, open;
  } chain_acc;

static int chain_close(targ *T, const chain_acc *C, const fga_unit *U, int isnew)
{ const fga_chain_params *P = T->prm;
  fga_hit H;
  int64_t dlo, dhi, alo = C->alow, ahi = C->reach;
  if (!C->open || C->cov < P->chain_min || (C->sides == 1 && !isnew))
    return 0;
  dlo = C->dgmin + (U->bucket << BUCK_SHIFT);
  dhi = C->dgmax + (U->bucket << BUCK_SHIFT);
  if (U->comp)                         /* back to contig coordinates (FastGA.c:3205-3216) */
    { const int64_t alen = P->alen[U->actg];
      dlo += alen - (P->amxpos + P->bmxpos); dhi += alen - (P->amxpos + P->bmxpos);
      alo += alen - P->amxpos;               ahi += alen - P->amxpos;
    }
  else
    { dlo -= P->bmxpos; dhi -= P->bmxpos; }
  H.dgmin = (int32_t) dlo; H.dgmax = (int32_t) dhi;
  H.alow = alo; H.ahgh = ahi;
  H.cov = (int32_t) C->cov; H.pad = 0;
  return push_hit(&T->out,&H);
}

static int unit_chains(targ *T, int64_t b, int64_t m, int64_t e, const fga_unit *U, int isnew)
{ const key128 *K = T->keys;
  const layout *L = &T->L;
  const int64_t gap = T->prm->chain_break;
  int64_t s = b, t = m;
  chain_acc C;
  memset(&C,0,sizeof(C));
  C.reach = -gap;
  while (s < m || t < e)
    { /* next record of the anti-diagonal merge */
      const int from2 = (s >= m) || (t < e && field(K+t,L->s_anti,L->wt) < field(K+s,L->s_anti,L->wt));
      const int64_t x = from2 ? t++ : s++;
      const int64_t anti = (int64_t) field(K+x,L->s_anti,L->wt);
      const int64_t span = 2 * (int64_t) field(K+x,0,6);
      const int     dg   = (int) field(K+x,6,6) + (from2 ? BUCK_WIDTH : 0);
      if (anti >= C.reach + gap)
        { if (chain_close(T,&C,U,isnew)) return 1;
          C.open = 1; C.alow = anti; C.cov = span; C.reach = anti + span;
          C.sides = from2 ? 2 : 1; C.dgmin = C.dgmax = dg;
        }
      else
        { const int64_t top = anti + span, floor_ = C.reach > anti ? C.reach : anti;
          if (top > floor_) C.cov += top - floor_;
          if (top > C.reach) C.reach = top;
          C.sides |= from2 ? 2 : 1;
          if (dg < C.dgmin) C.dgmin = dg;
          if (dg > C.dgmax) C.dgmax = dg;
        }
    }
  return chain_close(T,&C,U,isnew);
}

/* Scan the units whose first bucket starts inside [c0,c1).  Units are independent: a unit is the bucket run d
 * starting at a "bucket head" plus the directly following run when it is bucket d+1 of the same segment (aux);
 * `isnew` only needs to know whether the run just before the head is bucket d-1 of the same segment.  So any
 * thread can start at any bucket head, which gives far more parallel slack than whole (strand,A,B) segments. */
static int scan_range(targ *T, int64_t c0, int64_t c1)
{ const key128 *K = T->keys;
  const layout *L = &T->L;
  const int64_t n = T->n;
  int64_t b, m, e;

#define BUCK(x)  ((int64_t) field(K+(x),L->s_buck,L->wd))
#define SAMEBK(x,sid,bk) (segid(K+(x),L) == (sid) && BUCK(x) == (bk))

  /* first bucket head at or after c0 */
  b = c0;
  if (b > 0)
    { uint64_t sid = segid(K+(b-1),L);
      int64_t  bk  = BUCK(b-1);
      while (b < n && SAMEBK(b,sid,bk))
        b += 1;
    }

  while (b < c1 && b < n)
    { const uint64_t sid = segid(K+b,L);
      const int64_t cdiag = BUCK(b);
      int isnew, aux;
      /* run d = [b,m) */
      m = b+1;
      while (m < n && SAMEBK(m,sid,cdiag))
        m += 1;
      /* run d+1 = [m,e) if present */
      e = m;
      while (e < n && SAMEBK(e,sid,cdiag+1))
        e += 1;
      aux = (e > m);
      isnew = !(b > 0 && SAMEBK(b-1,sid,cdiag-1));

      if (isnew || aux)
        { fga_unit U;
          const int64_t first = T->out.nhit;
          U.comp = (int) field(K+b,L->s_strand,1);
          U.actg = (int) field(K+b,L->s_a,L->wa);
          U.bctg = (int) field(K+b,L->s_b,L->wb);
          U.bucket = cdiag;
          if (unit_chains(T,b,m,e,&U,isnew)) return 1;
          if (T->out.nhit > first)
            { U.nhits = (int32_t) (T->out.nhit - first);
              U.first_hit = first;
              if (push_unit(&T->out,&U)) return 1;
            }
        }
      b = m;      /* the next unit starts at the following bucket head */
    }
  return 0;
}

static void *chain_thread(void *arg)
{ targ *T = arg;
  if (scan_range(T,T->c0,T->c1))
    T->status = 1;
  return NULL;
}

int fga_chain_scan(const void *keys, int64_t n, int wa, int wb, int wd, int wt,
                   const fga_chain_params *prm, int nthreads, fga_hits **out)
{ fga_hits *R;
  targ *T;
  pthread_t *th;
  int t;
  int64_t nh = 0, nu = 0;

  *out = NULL;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if (n < 100000) nthreads = 1;
  else if (n / nthreads < 20000) nthreads = (int) (n/20000) + 1;
  R = calloc(1,sizeof(fga_hits));
  T = calloc(nthreads,sizeof(targ));
  th = calloc(nthreads,sizeof(pthread_t));
  if (R == NULL || T == NULL || th == NULL)
    { fga_set_error("out of memory");
      free(R); free(T); free(th);
      return 1;
    }
  for (t = 0; t < nthreads; t++)
    { T[t].keys = keys; T[t].n = n;
      T[t].c0 = (n*t)/nthreads; T[t].c1 = (n*(t+1))/nthreads;
      T[t].L.wa = wa; T[t].L.wb = wb; T[t].L.wd = wd; T[t].L.wt = wt;
      T[t].L.s_anti = 12; T[t].L.s_buck = 12+wt; T[t].L.s_b = 12+wt+wd; T[t].L.s_a = 12+wt+wd+wb;
      T[t].L.s_strand = 12+wt+wd+wb+wa;
      T[t].prm = prm;
    }
  for (t = 1; t < nthreads; t++)
    pthread_create(th+t,NULL,chain_thread,T+t);
  chain_thread(T);
  for (t = 1; t < nthreads; t++)
    pthread_join(th[t],NULL);

  for (t = 0; t < nthreads; t++)
    { if (T[t].status)
        { fga_set_error("out of memory in chain scan");
          for (t = 0; t < nthreads; t++) { free(T[t].out.hits); free(T[t].out.units); }
          free(R); free(T); free(th);
          return 1;
        }
      nh += T[t].out.nhit;
      nu += T[t].out.nunit;
    }
  R->nhits = nh; R->nunits = nu;
  R->hits  = malloc(sizeof(fga_hit)*(nh+1));
  R->units = malloc(sizeof(fga_unit)*(nu+1));
  nh = nu = 0;
  for (t = 0; t < nthreads; t++)
    { int64_t q;
      memcpy(R->hits+nh,T[t].out.hits,sizeof(fga_hit)*T[t].out.nhit);
      for (q = 0; q < T[t].out.nunit; q++)
        { R->units[nu+q] = T[t].out.units[q];
          R->units[nu+q].first_hit += nh;
        }
      nh += T[t].out.nhit;
      nu += T[t].out.nunit;
      free(T[t].out.hits); free(T[t].out.units);
    }
  free(T); free(th);
  *out = R;
  return 0;
}

int fga_hits_create(const fga_unit *units, int64_t nunits, const fga_hit *hits, int64_t nhits, fga_hits **out)
{ fga_hits *R = calloc(1,sizeof(fga_hits));
  *out = NULL;
  if (R == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  R->nhits = nhits; R->nunits = nunits;
  R->hits  = malloc(sizeof(fga_hit)*(nhits+1));
  R->units = malloc(sizeof(fga_unit)*(nunits+1));
  if (R->hits == NULL || R->units == NULL)
    { free(R->hits); free(R->units); free(R);
      fga_set_error("out of memory");
      return 1;
    }
  memcpy(R->hits,hits,sizeof(fga_hit)*nhits);
  memcpy(R->units,units,sizeof(fga_unit)*nunits);
  *out = R;
  return 0;
}

void fga_hits_free(fga_hits *H)
{ if (H == NULL) return;
  free(H->hits); free(H->units); free(H);
}

int64_t         fga_hits_count(const fga_hits *H)   { return H->nhits; }
int64_t         fga_hits_nunits(const fga_hits *H)  { return H->nunits; }
const fga_hit  *fga_hits_array(const fga_hits *H)   { return H->hits; }
const fga_unit *fga_hits_units(const fga_hits *H)   { return H->units; }
