/* fga_par.c -- the host side's parallel loops: a team of threads that lives for one stage (redundancy filter, chain
 * ordering, .1aln encoding) and runs that stage's O(n) passes -- gathers, key builds, copies, an LSD radix sort of
 * (key,value) pairs -- each split into contiguous slices.  A stage at 10^6 records is a dozen such passes; starting
 * threads for each one would cost more than the passes themselves, so the team waits on a condition variable between
 * passes.  With one thread (or small n) everything runs inline in the caller.
 *
 * Reference counterpart: the reference runs its sorts / merges on NTHREADS pthreads per phase (FastGA.c:3800-4133).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "fga_host.h"

struct fga_team
  { int              nthreads;
    pthread_t       *th;
    pthread_mutex_t  lock;
    pthread_cond_t   go, done;
    long             epoch;            /* bumped for every pass */
    int              pending;          /* workers still inside the current pass */
    int              quit;
    fga_slice_fn     fn;
    void            *arg;
    int64_t          n;
    struct team_worker { struct fga_team *team; int id; } *w;
  };

static void slice_of(int64_t n, int nthreads, int id, int64_t *beg, int64_t *end)
{ *beg = n * id / nthreads;
  *end = n * (id+1) / nthreads;
}

static void *team_worker_main(void *arg)
{ struct team_worker *W = arg;
  struct fga_team *T = W->team;
  long seen = 0;
  for (;;)
    { fga_slice_fn fn; void *a; int64_t n, b, e;
      pthread_mutex_lock(&T->lock);
      while (T->epoch == seen && !T->quit)
        pthread_cond_wait(&T->go,&T->lock);
      if (T->quit)
        { pthread_mutex_unlock(&T->lock);
          return NULL;
        }
      seen = T->epoch;
      fn = T->fn; a = T->arg; n = T->n;
      pthread_mutex_unlock(&T->lock);
      slice_of(n,T->nthreads,W->id,&b,&e);
      fn(a,W->id,b,e);
      pthread_mutex_lock(&T->lock);
      if (--T->pending == 0)
        pthread_cond_signal(&T->done);
      pthread_mutex_unlock(&T->lock);
    }
}

fga_team *fga_team_open(int nthreads)
{ fga_team *T = calloc(1,sizeof(fga_team));
  int t;
  if (T == NULL)
    return NULL;
  if (nthreads > 64) nthreads = 64;
  if (nthreads < 1) nthreads = 1;
  T->nthreads = 1;
  pthread_mutex_init(&T->lock,NULL);
  pthread_cond_init(&T->go,NULL);
  pthread_cond_init(&T->done,NULL);
  if (nthreads == 1)
    return T;
  T->th = calloc(nthreads,sizeof(pthread_t));
  T->w  = calloc(nthreads,sizeof(*T->w));
  if (T->th == NULL || T->w == NULL)
    { free(T->th); free(T->w); T->th = NULL; T->w = NULL;
      return T;                                    /* a team of one still works */
    }
  T->nthreads = nthreads;
  for (t = 1; t < nthreads; t++)
    { T->w[t].team = T; T->w[t].id = t;
      if (pthread_create(T->th+t,NULL,team_worker_main,T->w+t) != 0)
        { T->nthreads = t;                         /* the ones that started */
          break;
        }
    }
  return T;
}

int fga_team_size(const fga_team *T)
{ return T->nthreads; }

/* fn(arg, slice id, begin, end) over [0,n) cut into one contiguous slice per thread; returns when all are done */
void fga_team_run(fga_team *T, int64_t n, fga_slice_fn fn, void *arg)
{ int64_t b, e;
  if (T->nthreads == 1)
    { fn(arg,0,0,n);
      return;
    }
  pthread_mutex_lock(&T->lock);
  T->fn = fn; T->arg = arg; T->n = n;
  T->pending = T->nthreads - 1;
  T->epoch += 1;
  pthread_cond_broadcast(&T->go);
  pthread_mutex_unlock(&T->lock);
  slice_of(n,T->nthreads,0,&b,&e);
  fn(arg,0,b,e);
  pthread_mutex_lock(&T->lock);
  while (T->pending > 0)
    pthread_cond_wait(&T->done,&T->lock);
  pthread_mutex_unlock(&T->lock);
}

void fga_team_close(fga_team *T)
{ int t;
  if (T == NULL)
    return;
  if (T->nthreads > 1)
    { pthread_mutex_lock(&T->lock);
      T->quit = 1;
      pthread_cond_broadcast(&T->go);
      pthread_mutex_unlock(&T->lock);
      for (t = 1; t < T->nthreads; t++)
        pthread_join(T->th[t],NULL);
    }
  pthread_mutex_destroy(&T->lock);
  pthread_cond_destroy(&T->go);
  pthread_cond_destroy(&T->done);
  free(T->th); free(T->w); free(T);
}

/* ---- stable LSD radix sort of (key, value) pairs on the low `bits` bits of the key, 11 bits per pass ----
 * Every thread counts the digits of its slice, the counts are turned into start positions in (digit, thread) order,
 * and every thread scatters its slice: stable, and the same result for any number of threads. */
#define RDIG 2048

typedef struct
  { const uint64_t *ka; const int64_t *va;
    uint64_t *kb; int64_t *vb;
    int64_t  *cnt;                 /* [nthreads][RDIG] */
    int       shift;
  } sort_pass;

static void sort_count(void *arg, int id, int64_t b, int64_t e)
{ sort_pass *P = arg;
  int64_t *c = P->cnt + (int64_t) id*RDIG, i;
  memset(c,0,sizeof(int64_t)*RDIG);
  for (i = b; i < e; i++)
    c[(P->ka[i] >> P->shift) & (RDIG-1)] += 1;
}

static void sort_scatter(void *arg, int id, int64_t b, int64_t e)
{ sort_pass *P = arg;
  int64_t *c = P->cnt + (int64_t) id*RDIG, i;
  for (i = b; i < e; i++)
    { const int64_t d = c[(P->ka[i] >> P->shift) & (RDIG-1)]++;
      P->kb[d] = P->ka[i]; P->vb[d] = P->va[i];
    }
}

int fga_team_sort_pairs(fga_team *T, uint64_t *key, int64_t *val, int64_t n, int bits)
{ const int nt = T->nthreads;
  uint64_t *k2 = malloc(sizeof(uint64_t)*(n > 0 ? n : 1));
  int64_t  *v2 = malloc(sizeof(int64_t)*(n > 0 ? n : 1));
  int64_t  *cnt = malloc(sizeof(int64_t)*RDIG*nt);
  uint64_t *ka = key, *kb = k2;
  int64_t  *va = val, *vb = v2;
  int shift;
  if (k2 == NULL || v2 == NULL || cnt == NULL)
    { free(k2); free(v2); free(cnt);
      return 1;
    }
  for (shift = 0; shift < bits && n > 1; shift += 11)
    { sort_pass P;
      int64_t sum = 0, first;
      int d, t;
      P.ka = ka; P.va = va; P.kb = kb; P.vb = vb; P.cnt = cnt; P.shift = shift;
      fga_team_run(T,n,sort_count,&P);
      first = 0;
      for (t = 0; t < nt; t++)
        first += cnt[(int64_t) t*RDIG + ((ka[0] >> shift) & (RDIG-1))];
      if (first == n)
        continue;                                   /* this digit is the same everywhere */
      for (d = 0; d < RDIG; d++)
        for (t = 0; t < nt; t++)
          { const int64_t c = cnt[(int64_t) t*RDIG + d];
            cnt[(int64_t) t*RDIG + d] = sum;
            sum += c;
          }
      fga_team_run(T,n,sort_scatter,&P);
      { uint64_t *x = ka; ka = kb; kb = x; }
      { int64_t *x = va; va = vb; vb = x; }
    }
  if (ka != key)
    { memcpy(key,ka,sizeof(uint64_t)*n); memcpy(val,va,sizeof(int64_t)*n); }
  free(k2); free(v2); free(cnt);
  return 0;
}
