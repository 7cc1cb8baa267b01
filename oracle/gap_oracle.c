/* gap_oracle.c -- TEST INFRASTRUCTURE (oracle/): CPU restatement of Gap_Improver (reference align.c:6714-7133).
 *
 * Every reader of a .1aln that prints base-level alignments calls Compute_Trace_PTS and then Gap_Improver (ALNtoPAF.c:278-280,
 * ALNshow.c:524-526, ALNtoPSL.c:193-195): the second rewrites the indel list so that runs of nearby same-direction gaps are
 * re-solved with "one difference per gap whatever its length", i.e. fewer, longer gaps.  The list keeps its length; only the
 * positions of the indels inside a "box" move, and Path.diffs changes by the substitutions gained or lost.
 * tests/test_oracle_vs_reference.py pins this file call by call against the real reference (oracle/_ref/libalign_ref.so).
 * Nothing under fastga_amd/ may call it.
 *
 * Vocabulary.  An entry -p is a gap in A before its p-th base (B has an extra base), +q a gap in B before its q-th base; both
 * 1-based.  A box is a maximal run of entries of one sign whose successive positions are less than LONG_SNAKE apart.  Inside
 * a box the "primary" sequence P is the one the positions refer to (A for negative entries, B for positive), Q the other, and
 * on diagonal m the partner of P[i] is Q[i + sg*m] with sg = -1 (A boxes) or +1 (B boxes); a box walks the diagonals from
 * the one before its first gap to the one after its last in steps of sg.  The two mirror-image halves of the reference
 * routine (align.c:6817-6963 and 6965-7128) are this one routine under that substitution.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LONG_SNAKE 50        /* align.c:6606 */

static int run_fwd(const char *p, const char *q)                /* snake, align.c:6638-6650 */
{ int i;
  for (i = 0; ; i++)
    if (p[i] == 4 || p[i] != q[i])
      return i;
}

static int run_bwd(const char *p, const char *q)                /* rsnake, align.c:6652-6664 */
{ int i;
  for (i = 0; ; i++)
    if (p[-1-i] == 4 || p[-1-i] != q[-1-i])
      return i;
}

static int mismatches(const char *p, const char *q, int n)      /* hamming, align.c:6620-6636 */
{ int i, h = 0;
  for (i = 0; i < n; i++)
    { if (p[i] == 4 || q[i] == 4)
        break;
      if (p[i] != q[i])
        h += 1;
    }
  return h;
}

/* aseq/bseq point at base 0 of sequences with a 4 before the first and after the last base.
 * t[0..T) is the indel list from Compute_Trace_PTS and is rewritten in place; *diffs is adjusted.  Returns 0, 1 = no memory */
int oracle_gap_improver(const char *aseq, int alen, const char *bseq, int blen, int abpos, int bbpos,
                        int *t, int T, int *diffs)
{ const char *A = aseq-1, *B = bseq-1;                         /* 1-based, like the reference */
  int  *F = NULL, *H;
  long  have = 0;
  int   x = 0, d = abpos-bbpos, cdiff = 0;

  while (x < T)
    { /* ---- delimit the next box: entries [first,x), diagonal dfirst before it, d after it ---- */
      const int first = x, dfirst = d, sg = t[x] < 0 ? -1 : 1;
      const char *P = sg < 0 ? A : B, *Q = sg < 0 ? B : A;
      const int plen = sg < 0 ? alen : blen;
      int gaps = 0, hamm = 0, fpos, lpos;

      fpos = sg*t[x];
      for (;;)
        { const int pos = sg*t[x];
          int nxt;
          while (x < T && t[x] == sg*pos)                      /* one gap = a run of equal entries */
            { x += 1; d += sg; }
          gaps += 1;
          lpos = pos;
          if (x >= T || (t[x] < 0) != (sg < 0))
            break;
          nxt = sg*t[x];
          if (nxt-pos >= LONG_SNAKE)
            break;
          hamm += mismatches(P+pos,Q+(pos+sg*d),nxt-pos);
        }
      if (gaps == 1)
        continue;

      { const int ndiag = x-first+1;                           /* = |dfirst-d|+1 */
        const int budget = gaps+hamm;
        int bound, passes, reach, i, g;
        int *h;

        if ((long) ndiag*(budget+3) > have)
          { have = (long) ndiag*(budget+3) + 1024;
            free(F);
            F = malloc(sizeof(int)*have);
            if (F == NULL)
              return 1;
          }
        H = F+ndiag;

        /* grow the box over the mismatched columns next to it, but not across the neighbouring indels */
        if (first == 0)
          bound = 0;
        else
          { const int e = t[first-1];
            bound = (e < 0) == (sg < 0) ? sg*e : (e < 0 ? -e : e) - sg*dfirst;
          }
        while (P[fpos-1] != Q[fpos-1+sg*dfirst] && P[fpos-1] != 4 && Q[fpos-1+sg*dfirst] != 4)
          { if (fpos <= bound)
              break;
            fpos -= 1;
          }
        if (x >= T)
          bound = plen;
        else
          { const int e = t[x];
            bound = (e < 0) == (sg < 0) ? sg*e : (e < 0 ? -e : e) - sg*d;
          }
        while (P[lpos] != Q[lpos+sg*d] && P[lpos] != 4 && Q[lpos+sg*d] != 4)
          { if (lpos >= bound)
              break;
            lpos += 1;
          }

        /* furthest-reaching passes: F[i] = furthest position on the i-th diagonal of the box, H = the move taken:
         * 0 = a substitution on the same diagonal, c > 0 = a gap of c diagonals.  g is the tie-break counter: the
         * reference keeps it in the first word of its G vector (its cursor into G never advances, align.c:6868-6905, and
         * the word is not initialised), so it is one counter for the whole box, bumped each time a gap move is recorded,
         * and only differences of it are ever compared; it starts from 0 here. */
        F[0] = fpos + run_fwd(P+fpos,Q+(fpos+sg*dfirst));
        g = 0;
        for (i = 1; i < ndiag; i++)
          F[i] = fpos-2;
        passes = 0;
        h = H;
        reach = fpos;
        while (reach < lpos && passes < budget)
          { int best = fpos, c = 0, u = 0x7fffffff;
            for (i = 0; i < ndiag; i++)
              { const int m = dfirst + sg*i;
                int n = F[i], p;
                if (n >= best)
                  { p = n+1;
                    *h++ = 0;
                    if (n > best)
                      { c = 0; u = g+1; best = n; }
                    else if (g+1 < u)
                      { c = 0; u = g+1; }
                    else
                      c += 1;
                  }
                else
                  { n += 1;
                    p = best;
                    c += 1;
                    if (n == best && g < u)
                      *h++ = 0;
                    else
                      { *h++ = c; g = u; }
                  }
                p += run_fwd(P+p,Q+(p+sg*m));
                F[i] = p;
                reach = p;
              }
            passes += 1;
          }

        if (reach >= lpos && passes < budget)                  /* strictly fewer differences: rewrite the box */
          { int p = lpos, m = d, y = x, nham = 0, k;
            while (h > H)
              { p -= run_bwd(P+p,Q+(p+sg*m));
                if (p < fpos)
                  p = fpos;
                h -= ndiag;
                k = h[sg*(m-dfirst)];
                if (k == 0)
                  { p -= 1; nham += 1; }
                else
                  { m -= sg*k;
                    for (; k > 0; k--)
                      t[--y] = sg*p;
                  }
              }
            cdiff += nham-hamm;
          }
      }
    }

  free(F);
  *diffs += cdiff;
  return 0;
}
