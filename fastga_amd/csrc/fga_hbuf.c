/* fga_hbuf.c -- the library's large host arrays, kept between comparisons.
 *
 * One 3 Gbp comparison allocates, fills once and frees ~1.2 GB of host memory in blocks of tens to hundreds of MB (the raw
 * and the filtered record sets with their trace bytes, the formatted .1aln stretches, index arrays).  glibc serves such
 * blocks with mmap and gives them back with munmap, so EVERY comparison of a session pays the first touch of every page
 * again: 0.17 s of a 1.53 s run (measured by running the same binary with glibc's MALLOC_MMAP_MAX_=0
 * MALLOC_TRIM_THRESHOLD_=64G).  The reference never meets this: it is one process per comparison and streams its records
 * through files.  Changing the process's malloc policy is not a library's business, so blocks of FGA_BIG_MIN bytes and more
 * go through a small cache of their own: sizes rounded up to eight classes per octave, freed blocks parked (up to
 * FGA_HOST_CACHE_MB, default 16 GB, the least recently parked dropped first) and handed out again warm.  malloc / calloc /
 * realloc / free of the library's C sources are macros onto these (fga_host.h); everything smaller, and every pointer that did
 * not come from here (asprintf, getcwd, the caller's), takes the plain path -- told apart without a lock by the page
 * alignment of the big blocks.
 */
#define FGA_NO_MALLOC_MACROS
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <malloc.h>

#include "fga_host.h"

#define FGA_BIG_MIN   ((size_t) 4 << 20)
#define MAXBLOCKS     4096

typedef struct { void *p; size_t cap; int busy; unsigned long stamp; } hblock;

static pthread_mutex_t hmu = PTHREAD_MUTEX_INITIALIZER;
static hblock  hb[MAXBLOCKS];
static int     nhb = 0;
static size_t  parked = 0;                 /* bytes of the blocks that are not in use */
static unsigned long clockv = 0;

static size_t cache_cap(void)
{ static size_t cap = 0;
  if (cap == 0)
    { const char *e = getenv("FGA_HOST_CACHE_MB");
      cap = (e != NULL && atoll(e) >= 0) ? (size_t) atoll(e) << 20 : (size_t) 16 << 30;
      if (cap == 0) cap = 1;               /* (0: nothing is parked) */
    }
  return cap;
}

static size_t size_class(size_t n)         /* n rounded up to a multiple of 1/8 of its power of two */
{ size_t p = FGA_BIG_MIN, step;
  while (p < n && p < ((size_t) 1 << 62)) p <<= 1;
  step = p >> 4;                           /* p/2 < n <= p: classes p/2 + k * p/16 */
  return (n + step - 1) / step * step;
}

/* the least recently parked blocks go until the parked bytes fit (called with the lock held) */
static void shed(size_t limit)
{ while (parked > limit)
    { int i, best = -1;
      for (i = 0; i < nhb; i++)
        if (!hb[i].busy && (best < 0 || hb[i].stamp < hb[best].stamp))
          best = i;
      if (best < 0) break;
      parked -= hb[best].cap;
      free(hb[best].p);
      hb[best] = hb[--nhb];
    }
}

void *fga_big_malloc(size_t n)
{ size_t cap;
  void *p = NULL;
  int i, best = -1;
  if (n < FGA_BIG_MIN)
    return malloc(n);
  cap = size_class(n);
  pthread_mutex_lock(&hmu);
  for (i = 0; i < nhb; i++)
    if (!hb[i].busy && hb[i].cap == cap && (best < 0 || hb[i].stamp > hb[best].stamp))
      best = i;
  if (best >= 0)
    { hb[best].busy = 1;
      parked -= cap;
      p = hb[best].p;
    }
  pthread_mutex_unlock(&hmu);
  if (p != NULL)
    return p;
  if (posix_memalign(&p,4096,cap) != 0)
    { pthread_mutex_lock(&hmu);            /* give the parked memory back and try once more */
      shed(0);
      pthread_mutex_unlock(&hmu);
      if (posix_memalign(&p,4096,cap) != 0)
        return NULL;
    }
  pthread_mutex_lock(&hmu);
  if (nhb < MAXBLOCKS)
    { hb[nhb].p = p; hb[nhb].cap = cap; hb[nhb].busy = 1; hb[nhb].stamp = 0; nhb += 1; }
  /* (a full table: the block is a plain one, fga_big_free finds no entry and frees it) */
  pthread_mutex_unlock(&hmu);
  return p;
}

static int find_block(const void *p)       /* lock held */
{ int i;
  for (i = 0; i < nhb; i++)
    if (hb[i].p == p)
      return i;
  return -1;
}

void fga_big_free(void *p)
{ int i;
  if (p == NULL) return;
  if (((uintptr_t) p & 4095) != 0)         /* not one of the big blocks */
    { free(p);
      return;
    }
  pthread_mutex_lock(&hmu);
  i = find_block(p);
  if (i >= 0 && hb[i].busy)
    { hb[i].busy = 0;
      hb[i].stamp = ++clockv;
      parked += hb[i].cap;
      shed(cache_cap());
      pthread_mutex_unlock(&hmu);
      return;
    }
  pthread_mutex_unlock(&hmu);
  if (i < 0)
    free(p);                               /* a page-aligned pointer of someone else's */
  /* (i >= 0 and not busy: a second free of a parked block -- ignored) */
}

void *fga_big_calloc(size_t n, size_t m)
{ void *p;
  if (m != 0 && n > (size_t) -1 / m) return NULL;
  if (n*m < FGA_BIG_MIN)
    return calloc(n,m);
  p = fga_big_malloc(n*m);
  if (p != NULL) memset(p,0,n*m);
  return p;
}

void *fga_big_realloc(void *p, size_t n)
{ size_t have = 0;
  void *q;
  int i = -1;
  if (p == NULL) return fga_big_malloc(n);
  if (((uintptr_t) p & 4095) == 0)
    { pthread_mutex_lock(&hmu);
      i = find_block(p);
      if (i >= 0) have = hb[i].cap;
      pthread_mutex_unlock(&hmu);
    }
  if (i < 0)
    { if (n < FGA_BIG_MIN)
        return realloc(p,n);
      have = malloc_usable_size(p);        /* a plain block grows into a big one */
      q = fga_big_malloc(n);
      if (q == NULL) return NULL;
      memcpy(q,p,have < n ? have : n);
      free(p);
      return q;
    }
  if (n <= have)
    return p;
  q = fga_big_malloc(n);
  if (q == NULL) return NULL;
  memcpy(q,p,have);
  fga_big_free(p);
  return q;
}

/* the parked blocks back to the system (a session that is done for a while) */
void fga_host_cache_trim(void)
{ pthread_mutex_lock(&hmu);
  shed(0);
  pthread_mutex_unlock(&hmu);
}
