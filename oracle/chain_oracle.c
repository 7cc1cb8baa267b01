/* oracle/chain_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the chain scan of align_contigs (reference FastGA.c:3016-3176, 3340-3403; semantics in
 * SURVEY.md Appendix B.3) over sorted diagonal records given as plain field arrays, in the reference's own sequential
 * form: per (strand, A contig, B contig) run, bucket triples b,m,e (run d, run d+1), the anti-diagonal merge with its
 * flush step, chain break / coverage / mix / new rules, hit boxes shifted to contig coordinates (FastGA.c:3205-3216).
 * Pinned against the reference itself: tests/test_chain_oracle.py compares its boxes with the "Hit on bucket" /
 * "Box:" lines a DEBUG_HIT build of the reference prints (oracle/_ref/FastGA_hits, made by oracle/Makefile with sed in
 * a temp dir, FastGA.c:3165-3225).
 *
 * Input: n records sorted by (strand, actg, bctg, bucket, anti, ...) as seven int64 arrays.
 * Output: one row of 10 int64 per hit: strand, actg, bctg, bucket, aux, cov, dgmin, dgmax, alow, ahgh.
 */
#include <stdint.h>
#include <stdlib.h>

#define BUCK_WIDTH 64
#define BUCK_SHIFT 6

typedef struct
  { int64_t *rows; int64_t n, cap; } hitvec;

static int put(hitvec *v, const int64_t *row)
{ int k;
  if (v->n >= v->cap)
    { v->cap = 2*v->cap + 1024;
      v->rows = realloc(v->rows,sizeof(int64_t)*10*(size_t) v->cap);
      if (v->rows == NULL) return 1;
    }
  for (k = 0; k < 10; k++) v->rows[10*v->n+k] = row[k];
  v->n += 1;
  return 0;
}

int64_t oracle_chain_scan(int64_t n, const int64_t *strand, const int64_t *actg, const int64_t *bctg,
                          const int64_t *bucket, const int64_t *anti, const int64_t *drem, const int64_t *lcp,
                          int64_t chain_break, int64_t chain_min, int64_t amxpos, int64_t bmxpos,
                          const int64_t *alen /* by sorted A contig index */, int64_t **rows_out)
{ hitvec V = { NULL, 0, 0 };
  int64_t beg = 0;
  *rows_out = NULL;
  while (beg < n)
    { int64_t end = beg+1, b, m, e;
      int     isnew = 1;
      const int64_t comp = strand[beg];
      const int64_t doffset = alen[actg[beg]] - (amxpos + bmxpos), aoffset = alen[actg[beg]] - amxpos;
      while (end < n && strand[end] == comp && actg[end] == actg[beg] && bctg[end] == bctg[beg])
        end += 1;
      /* FastGA.c:3043-3050: first bucket run */
      b = beg;
      e = b;
      while (e < end && bucket[e] == bucket[b]) e += 1;
      for (;;)
        { const int64_t cdiag = bucket[b];
          int aux = 0;
          m = e;
          while (e < end && bucket[e] == cdiag+1) { e += 1; aux = 1; }        /* FastGA.c:3059-3065 */
          if (isnew || aux)
            { int64_t s = b, t = m;
              int64_t ipost = anti[s], apost = aux ? anti[t] : INT64_MAX;
              int64_t ahgh = -chain_break, alow = apost < ipost ? apost : ipost, an, cov = 0;
              int64_t dgmin = 2*BUCK_WIDTH, dgmax = 0, dg, lc;
              int     go = 1, mix = 0, wch;
              while (go)                                                       /* FastGA.c:3110-3176 */
                { if (apost < ipost)
                    { lc = lcp[t]; dg = drem[t] + BUCK_WIDTH; an = apost;
                      t += 1;
                      apost = (t >= e) ? INT64_MAX : anti[t];
                      wch = 2;
                    }
                  else
                    { an = ipost;
                      if (s < m) { lc = lcp[s]; dg = drem[s]; } else lc = dg = 0;
                      s += 1;
                      if (s >= m)
                        { if (s > m) go = 0; else ipost = INT64_MAX; }
                      else
                        ipost = anti[s];
                      wch = 1;
                    }
                  lc <<= 1;
                  if (an < ahgh + chain_break)
                    { const int64_t cps = an + lc;
                      if (cps > ahgh)
                        { cov += (an >= ahgh) ? lc : cps-ahgh;
                          ahgh = cps;
                        }
                      mix |= wch;
                      if (dg < dgmin) dgmin = dg; else if (dg > dgmax) dgmax = dg;
                    }
                  else
                    { if (cov >= chain_min && (mix != 1 || isnew))
                        { int64_t row[10];
                          row[0] = comp; row[1] = actg[beg]; row[2] = bctg[beg]; row[3] = cdiag; row[4] = aux;
                          row[5] = cov;
                          row[6] = dgmin + (cdiag << BUCK_SHIFT); row[7] = dgmax + (cdiag << BUCK_SHIFT);
                          row[8] = alow; row[9] = ahgh;
                          if (comp)
                            { row[6] += doffset; row[7] += doffset; row[8] += aoffset; row[9] += aoffset; }
                          else
                            { row[6] -= bmxpos; row[7] -= bmxpos; }
                          if (put(&V,row)) return -1;
                        }
                      if (go)
                        { cov = lc; ahgh = an + lc; mix = wch; alow = an; dgmin = dgmax = dg; }
                    }
                }
            }
          /* FastGA.c:3389-3402: next triple */
          if (e >= end)
            break;
          if (aux)
            { b = m; isnew = 0; }
          else
            { b = e; isnew = 1;
              while (e < end && bucket[e] == bucket[b]) e += 1;
            }
        }
      beg = end;
    }
  *rows_out = V.rows;
  return V.n;
}

void oracle_chain_free(int64_t *rows) { free(rows); }
