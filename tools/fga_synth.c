/* fga_synth.c -- fast synthetic genome pairs for the human-scale configurations (BASELINE.json configs[3] / [4]:
 * 3 Gbp vs 3 Gbp).  Test / bench infrastructure, not part of the product library: fastga_amd/synth.py's numpy
 * recipe needs ~90 s per Gbp; this one makes 2 x 3 Gbp in seconds on the host cores.
 *
 * Same recipe: i.i.d. ACGT contigs; repeat copies (families of 300 / 1000 / 3000 / 6000 bases, each copy diverged
 * 1-15 % from its family, either strand) over a given fraction of the bases; genome B = genome A after block
 * rearrangements (exponential blocks, mean 40 kbp, a fraction inverted / swapped) and point divergence (60 %
 * substitutions, 20 % 1-bp deletions, 20 % 1-bp insertions).  Every contig draws from its own generator seeded by
 * (seed, contig, stream), so the result does not depend on the number of threads.
 *
 * Built as fastga_amd/libfga_synth.so (fastga_amd/csrc/Makefile); binding: fastga_amd/synth.py::write_pair_fast.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix(uint64_t *x)
{ uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed, uint64_t a, uint64_t b)
{ uint64_t x = seed * 0x2545f4914f6cdd1dull + a * 0x9e3779b97f4a7c15ull + b * 0xd1b54a32d192ed03ull + 1;
  int i;
  for (i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next(rng_t *r)          /* xoshiro256** */
{ uint64_t *s = r->s;
  const uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
  return res;
}

static inline double rng_unit(rng_t *r) { return (double) (rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline int64_t rng_below(rng_t *r, int64_t n) { return (int64_t) (rng_unit(r) * (double) n); }

static void fill_random(rng_t *r, uint8_t *s, int64_t n)
{ int64_t i = 0;
  while (i + 32 <= n)
    { uint64_t x = rng_next(r);
      int k;
      for (k = 0; k < 32; k++, x >>= 2) s[i++] = (uint8_t) (x & 3);
    }
  if (i < n)
    { uint64_t x = rng_next(r);
      for (; i < n; i++, x >>= 2) s[i] = (uint8_t) (x & 3);
    }
}

static void revcomp_inplace(uint8_t *s, int64_t n)
{ int64_t i, j;
  for (i = 0, j = n - 1; i < j; i++, j--)
    { uint8_t a = (uint8_t) (3 - s[i]), b = (uint8_t) (3 - s[j]);
      s[i] = b; s[j] = a;
    }
  if (i == j) s[i] = (uint8_t) (3 - s[i]);
}

/* point divergence at total `rate` (geometric gaps between events); dst needs room for 2 n bases; returns its length */
static int64_t mutate(rng_t *r, const uint8_t *src, int64_t n, double rate, uint8_t *dst)
{ int64_t i = 0, o = 0;
  double lq;
  if (rate <= 0.)
    { memcpy(dst, src, (size_t) n);
      return n;
    }
  lq = log1p(-rate);
  while (i < n)
    { double u = rng_unit(r);
      int64_t gap = (u <= 0.) ? n : (int64_t) (log(u) / lq);
      double kind;
      if (gap > n - i) gap = n - i;
      memcpy(dst + o, src + i, (size_t) gap);
      o += gap; i += gap;
      if (i >= n) break;
      kind = rng_unit(r);
      if (kind < 0.6)
        dst[o++] = (uint8_t) ((src[i] + 1 + rng_below(r, 3)) & 3);
      else if (kind < 0.8)
        ;                                           /* deletion */
      else
        { dst[o++] = src[i];
          dst[o++] = (uint8_t) rng_below(r, 4);
        }
      i++;
    }
  return o;
}

typedef struct
  { uint64_t seed, bseed;
    int      ncontig, nfam;
    const int64_t *lens;
    double   divergence, repeat_frac, inv_frac, swap_frac, div_lo, div_hi;
    uint8_t **fam;
    int64_t *famlen;
    int      wantB;
    const char *prefA, *prefB;
    char   **textA, **textB;         /* formatted FASTA per contig */
    int64_t *tlenA, *tlenB;
    int64_t *blenB;
    pthread_mutex_t lock;
    int      next, failed;
  } job_t;

static const char LETTER[4] = { 'A', 'C', 'G', 'T' };

static char *format_fasta(const char *prefix, int c, const uint8_t *s, int64_t n, int64_t *tlen)
{ int64_t cap = n + n / 80 + 64, o;
  char *t = malloc((size_t) cap);
  int64_t i;
  if (t == NULL) return NULL;
  o = sprintf(t, ">%s%d\n", prefix, c);
  for (i = 0; i < n; i += 80)
    { int64_t m = n - i < 80 ? n - i : 80, k;
      for (k = 0; k < m; k++) t[o + k] = LETTER[s[i + k]];
      o += m;
      t[o++] = '\n';
    }
  *tlen = o;
  return t;
}

static int do_contig(job_t *J, int c)
{ const int64_t n = J->lens[c];
  uint8_t *a = malloc((size_t) n + 16), *scratch = malloc(2 * 6000 + 64);
  rng_t r;
  int ok = 0;
  if (a == NULL || scratch == NULL) goto done;
  rng_seed(&r, J->seed, (uint64_t) c, 1);
  fill_random(&r, a, n);
  if (J->repeat_frac > 0. && J->nfam > 0)
    { int64_t target = (int64_t) (J->repeat_frac * (double) n), placed = 0;
      rng_seed(&r, J->seed, (uint64_t) c, 2);
      while (placed < target)
        { int f = (int) rng_below(&r, J->nfam);
          double d = J->div_lo + (J->div_hi - J->div_lo) * rng_unit(&r);
          int64_t m = mutate(&r, J->fam[f], J->famlen[f], d, scratch), pos;
          if (rng_unit(&r) < 0.5) revcomp_inplace(scratch, m);
          if (n <= m + 1) break;
          pos = rng_below(&r, n - m);
          memcpy(a + pos, scratch, (size_t) m);
          placed += m;
        }
    }
  J->textA[c] = format_fasta(J->prefA, c, a, n, &J->tlenA[c]);
  if (J->textA[c] == NULL) goto done;
  if (J->wantB)
    { uint8_t *re = malloc((size_t) n + 16), *b = malloc(2 * (size_t) n + 64);
      int64_t *cut = NULL, nb = 0, cap = n / 1000 + 4, i, m;
      int *order = NULL;
      uint8_t *inv = NULL;
      if (re == NULL || b == NULL) { free(re); free(b); goto done; }
      rng_seed(&r, J->seed ^ (J->bseed * 0x632be59bd9b4e019ull), (uint64_t) c, 3);
      if (n >= 4000 && (J->inv_frac > 0. || J->swap_frac > 0.))
        { int64_t at = 0, o = 0, nsw;
          cut = malloc(sizeof(int64_t) * (size_t) (cap + 1));
          if (cut == NULL) { free(re); free(b); goto done; }
          cut[0] = 0;
          while (at < n)
            { double u = rng_unit(&r);
              int64_t len = (int64_t) (-40000. * log(u > 0. ? u : 1e-300));
              if (len < 1000) len = 1000;
              at += len;
              if (at > n) at = n;
              cut[++nb] = at;
            }
          order = malloc(sizeof(int) * (size_t) nb);
          inv = malloc((size_t) nb);
          if (order == NULL || inv == NULL) { free(re); free(b); free(cut); free(order); free(inv); goto done; }
          for (i = 0; i < nb; i++)
            { order[i] = (int) i;
              inv[i] = rng_unit(&r) < J->inv_frac;
            }
          nsw = (int64_t) (J->swap_frac * (double) nb / 2.);
          for (i = 0; i < nsw; i++)
            { int64_t x = rng_below(&r, nb), y = rng_below(&r, nb);
              int t = order[x]; order[x] = order[y]; order[y] = t;
            }
          for (i = 0; i < nb; i++)
            { int k = order[i];
              int64_t len = cut[k + 1] - cut[k];
              memcpy(re + o, a + cut[k], (size_t) len);
              if (inv[k]) revcomp_inplace(re + o, len);
              o += len;
            }
          free(cut); free(order); free(inv);
        }
      else
        memcpy(re, a, (size_t) n);
      m = mutate(&r, re, n, J->divergence, b);
      J->blenB[c] = m;
      J->textB[c] = format_fasta(J->prefB, c, b, m, &J->tlenB[c]);
      free(re); free(b);
      if (J->textB[c] == NULL) goto done;
    }
  ok = 1;
done:
  free(a); free(scratch);
  return ok;
}

static void *worker(void *arg)
{ job_t *J = arg;
  for (;;)
    { int c;
      pthread_mutex_lock(&J->lock);
      c = J->next++;
      pthread_mutex_unlock(&J->lock);
      if (c >= J->ncontig) break;
      if (!do_contig(J, c))
        { pthread_mutex_lock(&J->lock);
          J->failed = 1;
          pthread_mutex_unlock(&J->lock);
        }
    }
  return NULL;
}

static int write_all(const char *path, char **text, const int64_t *tlen, int n)
{ FILE *f = fopen(path, "wb");
  int c, ok = 1;
  if (f == NULL) return 0;
  for (c = 0; c < n; c++)
    if (fwrite(text[c], 1, (size_t) tlen[c], f) != (size_t) tlen[c]) ok = 0;
  if (fclose(f) != 0) ok = 0;
  return ok;
}

/* FASTA of genome A (contigs <prefA>0 .. of the given lengths, upper case) and, with fastaB != NULL, of genome B =
 * rearranged + diverged A (contigs <prefB>0 ..).  A depends on (seed, lens, repeat_frac, nfam) only; B also on bseed,
 * divergence, inv_frac, swap_frac -- so two B's of different divergence can be made for the same A.
 * blenB (may be NULL) receives B's contig lengths.  Returns 0 on success. */
int fga_synth_pair(uint64_t seed, uint64_t bseed, int ncontig, const int64_t *lens, double divergence,
                   double repeat_frac, int nfam, double inv_frac, double swap_frac,
                   const char *fastaA, const char *prefA, const char *fastaB, const char *prefB,
                   int64_t *blenB, int nthreads)
{ static const int64_t FAMLEN[4] = { 300, 1000, 3000, 6000 };
  job_t J;
  pthread_t *th = NULL;
  int i, rc = 1;

  memset(&J, 0, sizeof(J));
  J.seed = seed; J.bseed = bseed; J.ncontig = ncontig; J.lens = lens; J.divergence = divergence;
  J.repeat_frac = repeat_frac; J.nfam = nfam; J.inv_frac = inv_frac; J.swap_frac = swap_frac;
  J.div_lo = 0.01; J.div_hi = 0.15;
  J.wantB = fastaB != NULL; J.prefA = prefA; J.prefB = prefB ? prefB : "b";
  J.fam = calloc((size_t) (nfam > 0 ? nfam : 1), sizeof(uint8_t *));
  J.famlen = calloc((size_t) (nfam > 0 ? nfam : 1), sizeof(int64_t));
  J.textA = calloc((size_t) ncontig, sizeof(char *)); J.textB = calloc((size_t) ncontig, sizeof(char *));
  J.tlenA = calloc((size_t) ncontig, sizeof(int64_t)); J.tlenB = calloc((size_t) ncontig, sizeof(int64_t));
  J.blenB = calloc((size_t) ncontig, sizeof(int64_t));
  if (!J.fam || !J.famlen || !J.textA || !J.textB || !J.tlenA || !J.tlenB || !J.blenB) goto done;
  for (i = 0; i < nfam; i++)
    { rng_t r;
      rng_seed(&r, seed, (uint64_t) i, 0);
      J.famlen[i] = FAMLEN[i & 3];
      J.fam[i] = malloc((size_t) J.famlen[i]);
      if (J.fam[i] == NULL) goto done;
      fill_random(&r, J.fam[i], J.famlen[i]);
    }
  pthread_mutex_init(&J.lock, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > ncontig) nthreads = ncontig;
  th = malloc(sizeof(pthread_t) * (size_t) nthreads);
  if (th == NULL) goto done;
  for (i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &J);
  for (i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  if (J.failed) goto done;
  if (!write_all(fastaA, J.textA, J.tlenA, ncontig)) goto done;
  if (J.wantB && !write_all(fastaB, J.textB, J.tlenB, ncontig)) goto done;
  if (blenB != NULL) memcpy(blenB, J.blenB, sizeof(int64_t) * (size_t) ncontig);
  rc = 0;
done:
  if (J.fam) for (i = 0; i < nfam; i++) free(J.fam[i]);
  if (J.textA) for (i = 0; i < ncontig; i++) free(J.textA[i]);
  if (J.textB) for (i = 0; i < ncontig; i++) free(J.textB[i]);
  free(J.fam); free(J.famlen); free(J.textA); free(J.textB); free(J.tlenA); free(J.tlenB); free(J.blenB); free(th);
  return rc;
}
