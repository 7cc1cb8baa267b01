/* trace_oracle.c -- TEST INFRASTRUCTURE (oracle/): CPU restatement of Compute_Trace_PTS in its GREEDIEST mode.
 *
 * Restates Compute_Trace_PTS (reference align.c:6171-6308) and the O(NP) comparison it runs between successive trace
 * points, iter_np (align.c:5584-5903), as used by every caller in the reference (ALNtoPAF.c:278, ALNshow.c:524,
 * ALNtoPSL.c:193, ONEaln.c:1011: mode GREEDIEST, dlow = 1 > dhgh = -1, i.e. no band except the main diagonal of a self
 * comparison).  Given an alignment's trace points (per 100-base panel of A: #diffs, #bases of B) it rebuilds, panel by
 * panel, an edit script with the fewest differences and returns it in the daligner convention: one int per indel,
 * -(p) = dash in A before its p-th base, +(q) = dash in B before its q-th base (1-based, absolute in the sequence).
 * tests/test_oracle_vs_reference.py pins it call by call against the real reference (oracle/_ref/libalign_ref.so).
 *
 * Nothing under fastga_amd/ may call this; it is the checker for tests/.
 *
 * The comparison, per panel A[0..M) x B[0..N), del = M-N:  FV[d][k] = furthest B index reached on diagonal k (A index =
 * B index + k) with "cost" d, where a step along the diagonal over a mismatch costs 1, a step to the neighbouring diagonal
 * AWAY from del costs 2 (it must be paid back) and a step TOWARDS del is free; rows are swept Gauss-Seidel fashion from the
 * outside in (k = hgh..del+1 downwards, k = low..del-1 upwards, then k = del), so "towards del" moves read the row being
 * written.  FH[d][k] records the move: -1 / +1 = from row d-2, diagonal k-1 / k+1;  0 = from row d-1, same diagonal;
 * 2 / 4 = from the same row, diagonal k-1 / k+1;  3 = end of list.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct
  { int *fv, *fh;          /* (dmax+3) rows of `stride` ints each; row r holds cost d = r-2, column c diagonal k = c-koff */
    int  stride, koff;
  } np_rows;

#define FV(d,k) (R->fv[((d)+2)*R->stride + (k) + R->koff])
#define FH(d,k) (R->fh[((d)+2)*R->stride + (k) + R->koff])

/* one panel; returns the number of differences, -1 if more than dmax would be needed; appends the indels to *out */
static int panel_np(const char *A, int M, const char *B, int N, np_rows *R, int dmax, int posl, int posh,
                    int aoff, int boff, int **out)
{ const int del = M-N;
  int low = del < 0 ? del : 0, hgh = del < 0 ? 0 : del;
  int d, k;

  for (k = low-1; k <= hgh+1; k++)
    FV(-2,k) = FV(-1,k) = -2;
  FV(-1,0) = -1;
  low += 1;
  hgh -= 1;

  for (d = 0; ; d++)
    { int j;
      if (d > dmax)
        return -1;
      if ((d & 1) == 0)
        { if (low > posl) low -= 1;
          if (hgh < posh) hgh += 1;
        }
      FV(d,hgh+1) = FV(d,low-1) = -2;

#define SLIDE(kk)                                          \
      { const char *a = A + (kk);                          \
        const int lim = (N < M-(kk)) ? N : M-(kk);         \
        while (j < lim && B[j] == a[j])                    \
          j += 1;                                          \
        FV(d,kk) = j;                                      \
      }
#define CHOOSE(kk,am,ap,mcode,pcode)                       \
      { const int ac = FV(d-1,kk)+1;                       \
        if (ac < (am))                                     \
          { if ((ap) < (am)) { FH(d,kk) = (mcode); j = (am); } \
            else             { FH(d,kk) = (pcode); j = (ap); } \
          }                                                \
        else                                               \
          { if ((ap) < ac)   { FH(d,kk) = 0;       j = ac;   } \
            else             { FH(d,kk) = (pcode); j = (ap); } \
          }                                                \
      }

      j = -2;                                   /* above del: "towards del" = from k+1 of this very row */
      for (k = hgh; k > del; k--)
        { const int ap = j+1, am = FV(d-2,k-1);
          CHOOSE(k,am,ap,-1,4)
          SLIDE(k)
        }
      j = -2;                                   /* below del: "towards del" = from k-1 of this very row */
      for (k = low; k < del; k++)
        { const int ap = FV(d-2,k+1)+1, am = j;
          CHOOSE(k,am,ap,2,1)
          SLIDE(k)
        }
      { const int ap = FV(d,del+1)+1, am = j;    /* del itself: both neighbours of this row */
        CHOOSE(del,am,ap,2,4)
        SLIDE(del)
      }
      if (FV(d,del) >= N)
        break;
    }

  /* reverse the move list from (d,del) back to (0,0), then walk it forwards and emit the indels */
  { int e, h, m;
    FH(0,0) = 3;
    k = del;
    e = FH(d,k);
    FH(d,k) = 3;
    while (e != 3)
      { h = k+e;
        if (e > 1)       h -= 3;
        else if (e == 0) d -= 1;
        else             d -= 2;
        m = FH(d,h);
        FH(d,h) = e;
        e = m;
        k = h;
      }
    k = d = 0;
    e = FH(0,0);
    while (e != 3)
      { const int c = FV(d,k);
        h = k-e;
        if (e > 1)       h += 3;
        else if (e == 0) d += 1;
        else             d += 2;
        if (h > k)
          *(*out)++ = boff + 1 + c;
        else if (h < k)
          *(*out)++ = -(aoff + 1 + c + k);
        k = h;
        e = FH(d,h);
      }
  }
  return d + (del < 0 ? -del : del);
}

/* aseq/bseq: numeric (0..3) sequences of the two contigs (B already complemented for a complement alignment);
 * points: the alignment's trace as 16-bit pairs (diffs, b-bases), tlen entries; selfie: aseq and bseq are the same contig.
 * out must hold at least the sum of the diffs entries; returns 0, or 1 on an inconsistent trace. */
int oracle_trace_pts(const char *aseq, int alen, const char *bseq, int blen, int selfie,
                     int abpos, int bbpos, int aepos, int bepos, const uint16_t *points, int tlen, int tspace,
                     int *out, int *outlen, int *diffs_out)
{ np_rows R;
  int dmax = 0, nmax = 0, i, d, diffs = 0;
  int dlow = -0x3fffffff, dhgh = 0x3fffffff;
  int ab, ae, bb, be, db;
  int *o = out;

  for (i = 1; i < tlen; i += 2)
    { if (points[i-1] > dmax) dmax = points[i-1];
      if (points[i] > nmax)   nmax = points[i];
    }
  if (tlen <= 1)
    nmax = bepos-bbpos;
  if (dmax & 1)
    dmax += 1;
  R.stride = tspace + nmax + 3;
  R.koff   = nmax + 1;
  R.fv = malloc(sizeof(int)*(size_t) (dmax+3)*R.stride);
  R.fh = malloc(sizeof(int)*(size_t) (dmax+3)*R.stride);
  if (R.fv == NULL || R.fh == NULL)
    { free(R.fv); free(R.fh);
      return 1;
    }

  if (selfie)                       /* a self comparison stays on its side of the main diagonal (align.c:6258-6268) */
    { const int b0 = abpos-bbpos, e0 = aepos-bepos;
      if (b0 == 0 || e0 == 0 || (b0 > 0) != (e0 > 0))
        { free(R.fv); free(R.fh);
          return 1;
        }
      if (b0 < 0) dhgh = -1; else dlow = 1;
    }

  ab = abpos;
  ae = (ab/tspace)*tspace;
  bb = bbpos;
  db = ab-bb;
  for (i = 1; i < tlen-2; i += 2)
    { ae += tspace;
      be = bb + points[i];
      if (ae > alen || be > blen)
        goto bad;
      d = panel_np(aseq+ab,ae-ab,bseq+bb,be-bb,&R,dmax,dlow-db,dhgh-db,ab,bb,&o);
      if (d < 0)
        goto bad;
      diffs += d;
      ab = ae;
      bb = be;
      db = ab-bb;
    }
  ae = aepos;
  be = bepos;
  if (ae > alen || be > blen)
    goto bad;
  d = panel_np(aseq+ab,ae-ab,bseq+bb,be-bb,&R,dmax,dlow-db,dhgh-db,ab,bb,&o);
  if (d < 0)
    goto bad;
  diffs += d;

  free(R.fv); free(R.fh);
  *outlen = (int) (o-out);
  *diffs_out = diffs;
  return 0;

bad:
  free(R.fv); free(R.fh);
  return 1;
}
