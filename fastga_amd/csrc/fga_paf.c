/* fga_paf.c -- PAF emission for a finished alignment set (host side of the trace stage).
 *
 * What the reference does in a second process, ALNtoPAF (ALNtoPAF.c:103-636), for `FastGA -paf[x|m|s|S]`: per alignment
 * one PAF line; with a CIGAR or cs tag the indel list of Compute_Trace_PTS (here: fga_trace_pts, on the device) is first
 * regrouped into fewer, longer gaps by Gap_Improver (align.c:6714-7133) and then turned into run-length operations.
 * This file holds the gap regrouping (gap_regroup), the operation builder and the line formatter; alignments are
 * independent, so they are formatted by `nthreads` host threads into per-thread buffers written out in order.
 *
 * Sequence access mirrors ALNtoPAF.c:258-277 exactly, because the regrouping looks at the bases around an alignment:
 * A is the whole contig with the sentinel 4 before its first and after its last base, B is only the aligned piece
 * [bbpos,bepos) (reverse-complemented for a complement alignment) with the sentinel either side of the piece.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <pthread.h>
#include <sys/stat.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "fga_host.h"
#include "fastga_amd.h"

#define GAP_NEAR 50          /* LONG_SNAKE, align.c:6606: gaps closer than this belong to one box */

/* ------------------------------------------------------------------------------------------------------------------
 *  gap regrouping
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct
  { const uint8_t *P, *Q;    /* 1-based views: P = the sequence the positions of the box refer to, Q = the other  */
    int            sg;       /* -1: gaps in A (negative entries), +1: gaps in B                                   */
  } boxview;

static inline int same_run(const boxview *v, int p, int m)                   /* matching columns forward from p  */
{ const uint8_t *a = v->P+p, *b = v->Q+(p+v->sg*m);
  int i = 0;
  while (a[i] != 4 && a[i] == b[i])
    i += 1;
  return i;
}

static inline int same_run_back(const boxview *v, int p, int m)              /* matching columns backward from p */
{ const uint8_t *a = v->P+p, *b = v->Q+(p+v->sg*m);
  int i = 0;
  while (a[-1-i] != 4 && a[-1-i] == b[-1-i])
    i += 1;
  return i;
}

static inline int column_differs(const boxview *v, int p, int m)             /* a real mismatch, not a sentinel  */
{ const uint8_t x = v->P[p], y = v->Q[p+v->sg*m];
  return x != y && x != 4 && y != 4;
}

typedef struct
  { int  *buf;
    long  cap;
  } scratch;

/* where the regrouping of alignment i has to start: 0 for scripts as Compute_Trace_PTS leaves them, the entry the
 * device stopped at for scripts of fga_trace_pts_regrouped, -1 when the device finished the alignment */
static inline int todo_from(const fga_traces *tr, int64_t i)
{ return tr->resume == NULL ? 0 : tr->resume[i]; }

/* t[0..T): the indel list (-p: gap in A before its p-th base, +q: gap in B before its q-th base; 1-based).
 * A1/B1 are 1-based (A1[1] = first base).  `start` > 0: the entries before it have been regrouped already (on the
 * device, fga_trace_pts_regrouped: a box too large for a lane's scratch is handed back) and `start` is the first entry
 * of the next box.  Returns 0, or 1 when out of memory. */
static int gap_regroup(const uint8_t *A1, int alen, const uint8_t *B1, int blen, int abpos, int bbpos,
                       int32_t *t, int T, int start, int *diffs, scratch *S)
{ int x, diag = abpos-bbpos, gained = 0;

  for (x = 0; x < start; x++)
    diag += t[x] < 0 ? -1 : 1;

  while (x < T)
    { boxview v;
      const int first = x, d0 = diag;
      int ngap = 0, nsub = 0, lo, hi;

      v.sg = t[x] < 0 ? -1 : 1;
      v.P = v.sg < 0 ? A1 : B1;
      v.Q = v.sg < 0 ? B1 : A1;
      lo = hi = v.sg*t[x];
      while (1)                                   /* collect gaps of this sign while they stay near each other */
        { const int pos = v.sg*t[x];
          int nxt, i;
          while (x < T && t[x] == v.sg*pos)
            { x += 1; diag += v.sg; }
          ngap += 1;
          hi = pos;
          if (x >= T || (t[x] < 0) != (v.sg < 0))
            break;
          nxt = v.sg*t[x];
          if (nxt-pos >= GAP_NEAR)
            break;
          for (i = pos; i < nxt; i++)
            { const uint8_t a = v.P[i], b = v.Q[i+v.sg*diag];
              if (a == 4 || b == 4)
                break;
              nsub += (a != b);
            }
        }
      if (ngap < 2)
        continue;

      { const int nd = x-first+1, allowed = ngap+nsub;
        int *F, *H, *h;
        int lim, rounds = 0, far, tie = 0, i;

        if ((long) nd*(allowed+2) > S->cap)
          { S->cap = (long) nd*(allowed+2) + 4096;
            free(S->buf);
            S->buf = malloc(sizeof(int)*S->cap);
            if (S->buf == NULL)
              { S->cap = 0;
                return 1;
              }
          }
        F = S->buf;
        H = F+nd;

        /* widen the box over the mismatched columns touching it, never past the neighbouring indel */
        if (first == 0)
          lim = 0;
        else
          { const int e = t[first-1];
            lim = ((e < 0) == (v.sg < 0)) ? v.sg*e : abs(e) - v.sg*d0;
          }
        while (column_differs(&v,lo-1,d0) && lo > lim)
          lo -= 1;
        if (x >= T)
          lim = v.sg < 0 ? alen : blen;
        else
          { const int e = t[x];
            lim = ((e < 0) == (v.sg < 0)) ? v.sg*e : abs(e) - v.sg*diag;
          }
        while (column_differs(&v,hi,diag) && hi < lim)
          hi += 1;

        /* rounds of "one more difference", a gap of any length counting once; H keeps the move into every diagonal:
         * 0 = substitution, c = gap spanning c diagonals.  `tie` orders equally far moves as the reference does */
        F[0] = lo + same_run(&v,lo,d0);
        for (i = 1; i < nd; i++)
          F[i] = lo-2;
        h = H;
        far = lo;
        while (far < hi && rounds < allowed)
          { int lead = lo, span = 0, mark = 0x7fffffff;
            for (i = 0; i < nd; i++)
              { const int m = d0 + v.sg*i;
                int own = F[i], p;
                if (own >= lead)
                  { p = own+1;
                    *h++ = 0;
                    if (own > lead || tie+1 < mark)
                      { span = 0; mark = tie+1; lead = own; }
                    else
                      span += 1;
                  }
                else
                  { p = lead;
                    span += 1;
                    if (own+1 == lead && tie < mark)
                      *h++ = 0;
                    else
                      { *h++ = span; tie = mark; }
                  }
                far = F[i] = p + same_run(&v,p,m);
              }
            rounds += 1;
          }

        if (far >= hi && rounds < allowed)
          { int p = hi, m = diag, y = x, subs = 0;
            while (h > H)
              { int k;
                p -= same_run_back(&v,p,m);
                if (p < lo)
                  p = lo;
                h -= nd;
                k = h[v.sg*(m-d0)];
                if (k == 0)
                  { p -= 1; subs += 1; }
                else
                  { m -= v.sg*k;
                    while (k-- > 0)
                      t[--y] = v.sg*p;
                  }
              }
            gained += subs-nsub;
          }
      }
    }
  *diffs += gained;
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  sequence windows
 * ---------------------------------------------------------------------------------------------------------------- */

static inline int base_at(const fga_gdb *G, int c, int64_t i)
{ const uint8_t *src = G->bps + G->contigs[c].boff;
  return (src[i>>2] >> (2*(i&3))) & 3;
}

/* the four bases of a packed byte as four numeric bytes: in order, and reverse-complemented */
static uint32_t unpack_fwd[256], unpack_rc[256];

static void unpack_tables(void)
{ int b, q;
  for (b = 0; b < 256; b++)
    { uint8_t f[4], r[4];
      for (q = 0; q < 4; q++)
        { f[q] = (uint8_t) ((b >> (2*q)) & 3);
          r[3-q] = (uint8_t) (3 - f[q]);
        }
      memcpy(unpack_fwd+b,f,4);
      memcpy(unpack_rc+b,r,4);
    }
}

/* B piece [bb,be) of contig c in the alignment's orientation; out[0] and out[n+1] = 4; returns a pointer such that
 * ptr[q] is the q-th base (1-based) of the (complemented) contig for q in bb+1..be.  out needs n+8 bytes. */
static const uint8_t *load_piece(const fga_gdb *G, int c, int bb, int be, int comp, uint8_t *out)
{ const uint8_t *src = G->bps + G->contigs[c].boff;
  const int64_t len = G->contigs[c].clen;
  const int n = be-bb;
  int i = 0;
  out[0] = 4;
  if (comp)
    { int64_t q = len-1-bb;                          /* source position of out[1], walking down */
      for (; i < n && (q & 3) != 3; i++, q--)
        out[1+i] = (uint8_t) (3 - base_at(G,c,q));
      for (; i+4 <= n; i += 4, q -= 4)
        memcpy(out+1+i,unpack_rc + src[q>>2],4);
      for (; i < n; i++, q--)
        out[1+i] = (uint8_t) (3 - base_at(G,c,q));
    }
  else
    { int64_t q = bb;
      for (; i < n && (q & 3) != 0; i++, q++)
        out[1+i] = (uint8_t) base_at(G,c,q);
      for (; i+4 <= n; i += 4, q += 4)
        memcpy(out+1+i,unpack_fwd + src[q>>2],4);
      for (; i < n; i++, q++)
        out[1+i] = (uint8_t) base_at(G,c,q);
    }
  out[n+1] = 4;
  return out - bb;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  operations and text
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct
  { char *op;
    int  *ln;
    int   n, cap;
  } oplist;

typedef struct
  { char  *s;
    size_t n, cap;
  } text;

static inline int op_add(oplist *L, char op, int len)
{ if (len <= 0)
    return 0;
  if (L->n > 0 && L->op[L->n-1] == op)
    { L->ln[L->n-1] += len;
      return 0;
    }
  if (L->n >= L->cap)
    { int cap = L->cap*2 + 256;
      char *o = realloc(L->op,cap);
      int  *l;
      if (o == NULL) return 1;
      L->op = o;
      l = realloc(L->ln,sizeof(int)*cap);
      if (l == NULL) return 1;
      L->ln = l;
      L->cap = cap;
    }
  L->op[L->n] = op;
  L->ln[L->n++] = len;
  return 0;
}

static int tx_room(text *X, size_t need)
{ if (X->n+need > X->cap)
    { size_t cap = (X->n+need)*2 + 65536;
      char *s = realloc(X->s,cap);
      if (s == NULL) return 1;
      X->s = s; X->cap = cap;
    }
  return 0;
}

static inline void tx_char(text *X, char c) { X->s[X->n++] = c; }

static void tx_str(text *X, const char *s)
{ size_t l = strlen(s);
  memcpy(X->s+X->n,s,l);
  X->n += l;
}

/* a scaffold name as the reference's converters print it: up to the first white space (ALNtoPAF.c:763-783) */
static void tx_name(text *X, const char *s)
{ while (*s != '\0' && !(*s == ' ' || (*s >= '\t' && *s <= '\r')))
    tx_char(X,*s++);
}

static void tx_int(text *X, int64_t v)
{ char b[24];
  int  k = 0;
  if ((uint64_t) v < 100)                            /* run lengths: one or two digits almost always */
    { if (v >= 10)
        X->s[X->n++] = (char) ('0' + v/10);
      X->s[X->n++] = (char) ('0' + v%10);
      return;
    }
  if (v < 0)
    { tx_char(X,'-'); v = -v; }
  do { b[k++] = (char) ('0' + v%10); v /= 10; } while (v > 0);
  while (k > 0)
    tx_char(X,b[--k]);
}

/* columns k..k+len of A against h..h+len of B as one M, or as =/X runs.  With SSE2 the columns are compared sixteen at a
 * time into a bit mask of up to 64 columns and the runs are read off the mask (the loads may run up to 15 bytes past the
 * block: the sequence buffers carry that slack); the per-column loop of the other build is the same thing one at a time. */
static int op_block(oplist *L, const uint8_t *A1, const uint8_t *B1, int k, int h, int len, int eqx)
{ const uint8_t *a = A1+k, *b = B1+h;
  int i;
  if (!eqx)
    return op_add(L,'M',len);
#if defined(__SSE2__)
  for (i = 0; i < len; )
    { const int n = len-i < 64 ? len-i : 64;
      uint64_t eq = 0;
      int w, pos;
      for (w = 0; w < n; w += 16)
        { const __m128i x = _mm_loadu_si128((const __m128i *) (a+i+w)), y = _mm_loadu_si128((const __m128i *) (b+i+w));
          eq |= (uint64_t) (unsigned) _mm_movemask_epi8(_mm_cmpeq_epi8(x,y)) << w;
        }
      if (n < 64)
        eq &= (1ull << n) - 1;
      for (pos = 0; pos < n; )
        { const uint64_t rest = eq >> pos;
          int r;
          if (rest & 1)
            { r = ~rest == 0 ? 64 : __builtin_ctzll(~rest);
              if (op_add(L,'=',r)) return 1;
            }
          else
            { r = rest == 0 ? n-pos : __builtin_ctzll(rest);
              if (op_add(L,'X',r)) return 1;
            }
          pos += r;
        }
      i += n;
    }
#else
  for (i = 0; i < len; )
    { int j = i;
      while (j < len && a[j] == b[j])
        j += 1;
      if (op_add(L,'=',j-i)) return 1;
      i = j;
      while (j < len && a[j] != b[j])
        j += 1;
      if (op_add(L,'X',j-i)) return 1;
      i = j;
    }
#endif
  return 0;
}

/* the alignment as operations in A-forward order; *del = bases of B opposite gaps in A */
static int build_ops(oplist *L, const fga_aln *a, const int32_t *t, int T, const uint8_t *A1, const uint8_t *B1,
                     int eqx, int *del)
{ int k = a->abpos+1, h = a->bbpos+1, x, d = 0;
  L->n = 0;
  for (x = 0; x < T; x++)
    { const int e = t[x];
      int len;
      if (e < 0)
        { len = -e-k;
          if (op_block(L,A1,B1,k,h,len,eqx) || op_add(L,'D',1)) return 1;
          k += len; h += len+1;
          d += 1;
        }
      else
        { len = e-h;
          if (op_block(L,A1,B1,k,h,len,eqx) || op_add(L,'I',1)) return 1;
          k += len+1; h += len;
        }
    }
  if (op_block(L,A1,B1,k,h,a->aepos-k+1,eqx)) return 1;
  *del = d;
  return 0;
}

/* one chunk of consecutive alignments: formatted by whichever worker takes it, written by the caller's thread as soon
 * as every chunk before it has been written */
typedef struct
  { const fga_gdb *g1, *g2;
    const fga_alns *alns;
    const fga_traces *tr;
    int      flags;
    int      psl;            /* 1: PSL lines (ALNtoPSL.c:77-405) instead of PAF */
    int64_t  beg, end;
    text     out;
    int      status;
    int      done;
  } paf_job;

/* what a worker keeps from chunk to chunk */
typedef struct
  { oplist   L;
    scratch  S;
    uint8_t *abuf, *bbuf, *arev;
    int32_t *tcopy, *blk;
    int64_t  acap, bcap, tcap, kcap;
  } fmt_scratch;

static void fmt_free(fmt_scratch *W)
{ free(W->L.op); free(W->L.ln); free(W->S.buf); free(W->abuf); free(W->bbuf); free(W->arev); free(W->tcopy); free(W->blk);
  memset(W,0,sizeof(*W));
}

/* The bases of A an alignment with T indels can look at: Gap_Improver reads A and B column by column and stops at the
 * sentinels around the aligned piece of B (ALNtoPAF.c:258-277), so on A it stays within T+2 bases of [abpos,aepos] --
 * the diagonal moves by one per indel.  The window gets the sentinel where it ends at a contig end, like the reference's
 * whole-contig buffer; elsewhere its border is never consulted.  ptr[q] = q-th base (1-based), NULL when out of memory. */
static const uint8_t *load_a_window(const fga_gdb *G, int c, int abpos, int aepos, int T, fmt_scratch *W)
{ const int64_t len = G->contigs[c].clen;
  int64_t wb = (int64_t) abpos - T - 8, we = (int64_t) aepos + T + 8;
  if (wb < 0) wb = 0;
  if (we > len) we = len;
  if (we-wb+64 > W->acap)
    { W->acap = 2*(we-wb) + 4096;
      free(W->abuf);
      W->abuf = malloc(W->acap);
      if (W->abuf == NULL)
        { W->acap = 0;
          return NULL;
        }
    }
  return load_piece(G,c,(int) wb,(int) we,0,W->abuf);
}

static void put_divergence(text *X, const fga_aln *a, int64_t iid)
{ const int64_t len = a->aepos-a->abpos;
  const int64_t v = 10000 + (len > 0 ? (10000ll*(len-iid))/len : 0);      /* an empty A interval: dv 0 */
  tx_str(X,"\tdv:f:0.");
  tx_char(X,(char) ('0'+(v/1000)%10));
  tx_char(X,(char) ('0'+(v/100)%10));
  tx_char(X,(char) ('0'+(v/10)%10));
  tx_char(X,(char) ('0'+v%10));
}

static int format_paf(paf_job *J, fmt_scratch *W)
{ const fga_gdb *g1 = J->g1, *g2 = J->g2;
  const int cigar_m = (J->flags & FGA_PAF_CIGAR_M) != 0, cigar_x = (J->flags & FGA_PAF_CIGAR_X) != 0;
  const int cs_s = (J->flags & FGA_PAF_CS_SHORT) != 0, cs_l = (J->flags & FGA_PAF_CS_LONG) != 0;
  const int swap = (J->flags & FGA_PAF_SWAP) != 0;
  const int cigar = cigar_m || cigar_x, cs = cs_s || cs_l, bases = cigar || cs;
  const int eqx = !(cigar_m && !cs);                       /* =/X runs wanted: the columns have to be compared */
  static const char lower[4] = { 'a','c','g','t' }, upper[4] = { 'A','C','G','T' };
  text    *X = &J->out;
  oplist  *L = &W->L;
  int64_t  i;

  for (i = J->beg; i < J->end; i++)
    { const fga_aln *a = J->alns->alns+i;
      const int comp = (a->flags & 0x1) != 0;
      const fga_contig *ca = g1->contigs+a->aread, *cb = g2->contigs+a->bread;
      const fga_scaffold *sa = g1->scaffolds+ca->scaf, *sb = g2->scaffolds+cb->scaf;
      const char *na = g1->headers+sa->hoff, *nb = g2->headers+sb->hoff;
      int64_t bs, be;

      if (tx_room(X,strlen(na)+strlen(nb)+256)) goto oom;
      if (comp)
        { bs = cb->sbeg+cb->clen-a->bepos; be = cb->sbeg+cb->clen-a->bbpos; }
      else
        { bs = cb->sbeg+a->bbpos; be = cb->sbeg+a->bepos; }
      if (swap)
        { tx_name(X,nb); tx_char(X,'\t'); tx_int(X,sb->slen); tx_char(X,'\t'); tx_int(X,bs); tx_char(X,'\t'); tx_int(X,be);
          tx_char(X,'\t'); tx_char(X,comp ? '-' : '+'); tx_char(X,'\t');
          tx_name(X,na); tx_char(X,'\t'); tx_int(X,sa->slen); tx_char(X,'\t'); tx_int(X,ca->sbeg+a->abpos);
          tx_char(X,'\t'); tx_int(X,ca->sbeg+a->aepos);
        }
      else
        { tx_name(X,na); tx_char(X,'\t'); tx_int(X,sa->slen); tx_char(X,'\t'); tx_int(X,ca->sbeg+a->abpos);
          tx_char(X,'\t'); tx_int(X,ca->sbeg+a->aepos);
          tx_char(X,'\t'); tx_char(X,comp ? '-' : '+'); tx_char(X,'\t');
          tx_name(X,nb); tx_char(X,'\t'); tx_int(X,sb->slen); tx_char(X,'\t'); tx_int(X,bs); tx_char(X,'\t'); tx_int(X,be);
        }

      if (!bases)                     /* without base-level work the two counts are estimates (ALNtoPAF.c:596-621) */
        { const int64_t sum = (a->aepos-a->abpos) + (a->bepos-a->bbpos), iid = (sum-a->diffs)/2;
          tx_char(X,'\t'); tx_int(X,iid); tx_char(X,'\t'); tx_int(X,sum/2); tx_str(X,"\t255");
          put_divergence(X,a,iid);
          tx_str(X,"\tdf:i:"); tx_int(X,a->diffs);
          tx_char(X,'\n');
          continue;
        }

      { const int T = J->tr->tlen[i], blen = (int) cb->clen, n = a->bepos-a->bbpos;
        const int todo = todo_from(J->tr,i);
        int diffs = J->tr->diffs[i], del = 0, q, j, from, to, step;
        const uint8_t *A1 = NULL, *B1 = NULL, *Aw, *Bw;
        const int32_t *t = J->tr->trace+J->tr->toff[i];
        int64_t block, iid;

        if (todo >= 0 || eqx)                              /* the bases are needed to regroup and to tell = from X */
          { A1 = load_a_window(g1,a->aread,a->abpos,a->aepos,T,W);
            if (A1 == NULL) goto oom;
            if (n+64 > W->bcap)
              { W->bcap = 2*(int64_t) n + 4096;
                free(W->bbuf);
                W->bbuf = malloc(W->bcap);
                if (W->bbuf == NULL) { W->bcap = 0; goto oom; }
              }
            B1 = load_piece(g2,a->bread,a->bbpos,a->bepos,comp,W->bbuf);
          }
        if (todo >= 0)
          { if (T+1 > W->tcap)
              { W->tcap = 2*(int64_t) T + 1024;
                free(W->tcopy);
                W->tcopy = malloc(sizeof(int32_t)*W->tcap);
                if (W->tcopy == NULL) { W->tcap = 0; goto oom; }
              }
            memcpy(W->tcopy,t,sizeof(int32_t)*T);
            if (gap_regroup(A1,(int) ca->clen,B1,blen,a->abpos,a->bbpos,W->tcopy,T,todo,&diffs,&W->S)) goto oom;
            t = W->tcopy;
          }
        if (build_ops(L,a,t,T,A1,B1,eqx,&del)) goto oom;

        block = (a->aepos-a->abpos) + del;
        iid = block-diffs;
        tx_char(X,'\t'); tx_int(X,iid); tx_char(X,'\t'); tx_int(X,block); tx_str(X,"\t255");
        put_divergence(X,a,iid);
        tx_str(X,"\tdf:i:"); tx_int(X,diffs);

        if (swap)
          for (q = 0; q < L->n; q++)
            L->op[q] = L->op[q] == 'I' ? 'D' : (L->op[q] == 'D' ? 'I' : L->op[q]);
        from = 0; to = L->n; step = 1;
        if (comp && !swap)
          { from = L->n-1; to = -1; step = -1; }

        if (cigar)
          { if (tx_room(X,(size_t) L->n*12+64)) goto oom;
            tx_str(X,"\tcg:Z:");
            if (cigar_m && cs)                         /* =/X runs were built for the cs tag: print them merged */
              { int run = 0;
                for (q = from; q != to; q += step)
                  if (L->op[q] == 'I' || L->op[q] == 'D')
                    { if (run > 0) { tx_int(X,run); tx_char(X,'M'); run = 0; }
                      tx_int(X,L->ln[q]); tx_char(X,L->op[q]);
                    }
                  else
                    run += L->ln[q];
                if (run > 0) { tx_int(X,run); tx_char(X,'M'); }
              }
            else
              for (q = from; q != to; q += step)
                { tx_int(X,L->ln[q]); tx_char(X,L->op[q]); }
          }

        if (cs)
          { const int alen_seg = a->aepos-a->abpos;
            Aw = A1+a->abpos+1;
            Bw = B1+a->bbpos+1;
            if (comp && !swap)                         /* read both segments on the other strand, operations reversed */
              { uint8_t *r;
                free(W->arev);
                W->arev = malloc((size_t) alen_seg + n + 8);
                if (W->arev == NULL) goto oom;
                r = W->arev;
                for (j = 0; j < alen_seg; j++) r[j] = (uint8_t) (3-Aw[alen_seg-1-j]);
                for (j = 0; j < n; j++) r[alen_seg+j] = (uint8_t) (3-Bw[n-1-j]);
                Aw = r; Bw = r+alen_seg;
              }
            if (swap)
              { const uint8_t *c = Aw; Aw = Bw; Bw = c; }
            if (tx_room(X,(size_t) 3*(alen_seg+n) + (size_t) L->n*12 + 64)) goto oom;
            tx_str(X,"\tcs:Z:");
            for (q = from; q != to; q += step)
              { const int l = L->ln[q];
                switch (L->op[q])
                  { case '=':
                      if (cs_s)
                        { tx_char(X,':'); tx_int(X,l); }
                      else
                        { tx_char(X,'=');
                          for (j = 0; j < l; j++) tx_char(X,upper[Aw[j]]);
                        }
                      Aw += l; Bw += l;
                      break;
                    case 'X':
                      for (j = 0; j < l; j++)
                        { tx_char(X,'*'); tx_char(X,lower[Bw[j]]); tx_char(X,lower[Aw[j]]); }
                      Aw += l; Bw += l;
                      break;
                    case 'I':
                      tx_char(X,'+');
                      for (j = 0; j < l; j++) tx_char(X,lower[Aw[j]]);
                      Aw += l;
                      break;
                    case 'D':
                      tx_char(X,'-');
                      for (j = 0; j < l; j++) tx_char(X,lower[Bw[j]]);
                      Bw += l;
                      break;
                    default:
                      break;
                  }
              }
          }
        if (tx_room(X,8)) goto oom;
        tx_char(X,'\n');
      }
    }
  return 0;

oom:
  fga_set_error("fga_write_paf: out of memory");
  return 1;
}


/* One PSL line per alignment (gen_psl, ALNtoPSL.c:77-405): counts, strand, names and ranges, then the ungapped blocks.
 * Always base-level: the edit script is regrouped first, trailing indels at the very end are trimmed off. */
static int format_psl(paf_job *J, fmt_scratch *W)
{ const fga_gdb *g1 = J->g1, *g2 = J->g2;
  text    *X = &J->out;
  int64_t  i;

  for (i = J->beg; i < J->end; i++)
    { fga_aln a = J->alns->alns[i];                  /* a copy: the trim moves aepos / bepos */
      const int comp = (a.flags & 0x1) != 0;
      const fga_contig *ca = g1->contigs+a.aread, *cb = g2->contigs+a.bread;
      const fga_scaffold *sa = g1->scaffolds+ca->scaf, *sb = g2->scaffolds+cb->scaf;
      const char *na = g1->headers+sa->hoff, *nb = g2->headers+sb->hoff;
      const int n = a.bepos-a.bbpos;
      int T = J->tr->tlen[i], diffs = J->tr->diffs[i];
      int x, k, h, cut, nblk, prev;
      int ngapA = 0, ngapB = 0, runA = 0, runB = 0, subs, same;
      const int todo = todo_from(J->tr,i);
      const int32_t *tcopy = J->tr->trace+J->tr->toff[i];
      int32_t *blk;
      int64_t boff;

      if (todo >= 0)                                 /* the blocks need no bases, only an unfinished regrouping does */
        { const uint8_t *A1 = load_a_window(g1,a.aread,a.abpos,a.aepos,T,W), *B1;
          if (A1 == NULL) goto oom;
          if (n+64 > W->bcap)
            { W->bcap = 2*(int64_t) n + 4096;
              free(W->bbuf);
              W->bbuf = malloc(W->bcap);
              if (W->bbuf == NULL) { W->bcap = 0; goto oom; }
            }
          B1 = load_piece(g2,a.bread,a.bbpos,a.bepos,comp,W->bbuf);
          if (T+1 > W->tcap)
            { W->tcap = 2*(int64_t) T + 1024;
              free(W->tcopy);
              W->tcopy = malloc(sizeof(int32_t)*W->tcap);
              if (W->tcopy == NULL) { W->tcap = 0; goto oom; }
            }
          memcpy(W->tcopy,tcopy,sizeof(int32_t)*T);
          if (gap_regroup(A1,(int) ca->clen,B1,(int) cb->clen,a.abpos,a.bbpos,W->tcopy,T,todo,&diffs,&W->S)) goto oom;
          tcopy = W->tcopy;
        }

      for (cut = 0; T > 0 && tcopy[T-1] == -a.aepos-1; T--)       /* gaps after the last base of A */
        cut += 1;
      a.bepos -= cut; diffs -= cut;
      for (cut = 0; T > 0 && tcopy[T-1] == a.bepos+1; T--)        /* gaps after the last base of B */
        cut += 1;
      a.aepos -= cut; diffs -= cut;

      for (x = 0, prev = 0; x < T; prev = tcopy[x++])
        if (tcopy[x] < 0)
          { ngapA += 1; runA += (tcopy[x] != prev); }
        else
          { ngapB += 1; runB += (tcopy[x] != prev); }
      subs = diffs-(ngapA+ngapB);
      same = (a.aepos-a.abpos)-ngapB-subs;

      if (T+2 > W->kcap)
        { W->kcap = 2*(int64_t) T + 1024;
          free(W->blk);
          W->blk = malloc(sizeof(int32_t)*3*W->kcap);
          if (W->blk == NULL) { W->kcap = 0; goto oom; }
        }
      blk = W->blk;
      nblk = 0;                                      /* blocks as (length, A start, B start), 0-based */
      k = a.abpos+1; h = a.bbpos+1;
      for (x = 0; x <= T; x++)
        { int len;
          if (x == T)
            len = a.aepos-k+1;
          else if (tcopy[x] < 0)
            len = -tcopy[x]-k;
          else
            len = tcopy[x]-h;
          if (len > 0)
            { blk[3*nblk] = len; blk[3*nblk+1] = k-1; blk[3*nblk+2] = h-1;
              nblk += 1;
            }
          if (x < T)
            { if (tcopy[x] < 0) { k += len; h += len+1; }
              else              { k += len+1; h += len; }
            }
        }

      if (tx_room(X,strlen(na)+strlen(nb)+512+(size_t) nblk*40)) goto oom;
      tx_int(X,same); tx_char(X,'\t'); tx_int(X,subs); tx_str(X,"\t0\t0\t");
      tx_int(X,runB); tx_char(X,'\t'); tx_int(X,ngapB); tx_char(X,'\t');
      tx_int(X,runA); tx_char(X,'\t'); tx_int(X,ngapA); tx_char(X,'\t');
      tx_char(X,comp ? '-' : '+'); tx_char(X,'\t');
      tx_name(X,na); tx_char(X,'\t'); tx_int(X,sa->slen); tx_char(X,'\t');
      tx_int(X,ca->sbeg+a.abpos); tx_char(X,'\t'); tx_int(X,ca->sbeg+a.aepos); tx_char(X,'\t');
      tx_name(X,nb); tx_char(X,'\t'); tx_int(X,sb->slen); tx_char(X,'\t');
      if (comp)
        { boff = cb->sbeg+cb->clen;
          tx_int(X,boff-a.bepos); tx_char(X,'\t'); tx_int(X,boff-a.bbpos);
        }
      else
        { boff = cb->sbeg;
          tx_int(X,boff+a.bbpos); tx_char(X,'\t'); tx_int(X,boff+a.bepos);
        }
      tx_char(X,'\t'); tx_int(X,nblk); tx_char(X,'\t');
      for (x = 0; x < nblk; x++)
        { const int32_t *b = blk + 3*(comp ? nblk-1-x : x);
          tx_int(X,b[0]); tx_char(X,',');
        }
      tx_char(X,'\t');
      for (x = 0; x < nblk; x++)
        { const int32_t *b = blk + 3*(comp ? nblk-1-x : x);
          tx_int(X,comp ? sa->slen-(ca->sbeg+b[1]+b[0]) : ca->sbeg+b[1]); tx_char(X,',');
        }
      tx_char(X,'\t');
      for (x = 0; x < nblk; x++)
        { const int32_t *b = blk + 3*(comp ? nblk-1-x : x);
          tx_int(X,comp ? boff-(b[2]+b[0]) : boff+b[2]); tx_char(X,',');
        }
      tx_char(X,'\n');
    }
  return 0;

oom:
  fga_set_error("fga_write_psl: out of memory");
  return 1;
}

/* every record must lie inside the contigs it names (the sets come from arbitrary .1aln files through bin/ALNtoPAF) */
static int check_records(const char *who, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns)
{ int64_t i;
  for (i = 0; i < alns->naln; i++)
    { const fga_aln *a = alns->alns+i;
      if (a->aread < 0 || a->aread >= g1->ncontig || a->bread < 0 || a->bread >= g2->ncontig ||
          a->abpos < 0 || a->aepos < a->abpos || a->aepos > g1->contigs[a->aread].clen ||
          a->bbpos < 0 || a->bepos < a->bbpos || a->bepos > g2->contigs[a->bread].clen)
        { fga_set_error("%s: alignment %lld lies outside the contigs of the two genomes",who,(long long) i);
          return 1;
        }
    }
  return 0;
}

/* the formatting of a set: workers take chunks of consecutive alignments from a counter, the caller's thread writes
 * every finished chunk as soon as all chunks before it are out -- the file is written while the rest is formatted, and
 * a written chunk's text is given back */
typedef struct
  { paf_job        *chunk;
    int             nchunk, next, failed;
    pthread_mutex_t mu;
    pthread_cond_t  cv;
  } paf_run;

static void *paf_worker(void *arg)
{ paf_run    *R = arg;
  fmt_scratch W;
  memset(&W,0,sizeof(W));
  while (1)
    { const int c = __sync_fetch_and_add(&R->next,1);
      paf_job *J;
      int st;
      if (c >= R->nchunk || R->failed)
        break;
      J = R->chunk+c;
      st = J->psl ? format_psl(J,&W) : format_paf(J,&W);
      pthread_mutex_lock(&R->mu);
      J->status = st;
      J->done = 1;
      if (st) R->failed = 1;
      pthread_cond_broadcast(&R->cv);
      pthread_mutex_unlock(&R->mu);
    }
  fmt_free(&W);
  return NULL;
}

static int write_lines(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns,
                       const fga_traces *traces, int flags, int nthreads, int psl)
{ const int bases = psl || (flags & (FGA_PAF_CIGAR_M|FGA_PAF_CIGAR_X|FGA_PAF_CS_SHORT|FGA_PAF_CS_LONG)) != 0;
  const int to_stdout = (path == NULL || strcmp(path,"-") == 0);
  const char *who = psl ? "fga_write_psl" : "fga_write_paf";
  paf_run    R;
  pthread_t *th;
  int *tstarted;
  FILE      *f;
  int        t, c, nchunk, rc = 0, started = 0;
  int64_t    total = 0, acc = 0, i, nxt;
  double     t0, twait = 0., tfile = 0.;

  if (g2 == NULL) g2 = g1;
  if ((flags & FGA_PAF_CIGAR_M) && (flags & FGA_PAF_CIGAR_X))
    { fga_set_error("fga_write_paf: only one of the M and the X/= CIGAR forms can be asked for");
      return 1;
    }
  if ((flags & FGA_PAF_CS_SHORT) && (flags & FGA_PAF_CS_LONG))
    { fga_set_error("fga_write_paf: only one of the short and the long cs forms can be asked for");
      return 1;
    }
  if (bases && (traces == NULL || traces->naln != alns->naln))
    { fga_set_error("%s: base-level output needs the edit scripts of fga_trace_pts for the same alignments",who);
      return 1;
    }
  if (check_records(who,g1,g2,alns))
    return 1;
  unpack_tables();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  if (alns->naln < nthreads) nthreads = alns->naln > 0 ? (int) alns->naln : 1;
  nchunk = 8*nthreads;
  if (alns->naln < nchunk) nchunk = alns->naln > 0 ? (int) alns->naln : 1;
  memset(&R,0,sizeof(R));
  R.chunk = calloc(nchunk,sizeof(paf_job));
  th = calloc(nthreads,sizeof(pthread_t));
  tstarted = calloc(nthreads > 0 ? nthreads : 1,sizeof(int));
  if (R.chunk == NULL || th == NULL || tstarted == NULL)
    { free(R.chunk); free(th); free(tstarted);
      fga_set_error("out of memory");
      return 1;
    }
  R.nchunk = nchunk;
  for (i = 0; i < alns->naln; i++)                   /* equal shares of aligned bases, contiguous in file order */
    total += alns->alns[i].aepos-alns->alns[i].abpos + 200;
  nxt = 0;
  for (c = 0, i = 0; c < nchunk; c++)
    { paf_job *J = R.chunk+c;
      J->g1 = g1; J->g2 = g2; J->alns = alns; J->tr = traces; J->flags = flags; J->psl = psl;
      J->beg = i;
      nxt += total/nchunk + 1;
      while (i < alns->naln && (acc < nxt || c == nchunk-1))
        { acc += alns->alns[i].aepos-alns->alns[i].abpos + 200; i++; }
      J->end = i;
    }
  f = to_stdout ? stdout : fopen(path,"w");
  if (f == NULL)
    { fga_set_error("cannot create %s",path);
      free(R.chunk); free(th); free(tstarted);
      return 1;
    }
  pthread_mutex_init(&R.mu,NULL);
  pthread_cond_init(&R.cv,NULL);
  t0 = fga_wall();
  for (t = 0; t < nthreads; t++)
    { tstarted[t] = pthread_create(th+t,NULL,paf_worker,&R) == 0;
      if (tstarted[t]) started = 1;
    }
  if (!started)                                      /* no thread could be made: format here */
    paf_worker(&R);
  for (c = 0; c < nchunk && rc == 0; c++)
    { paf_job *J = R.chunk+c;
      const double w0 = fga_wall();
      double w1;
      pthread_mutex_lock(&R.mu);
      while (!J->done && !R.failed)
        pthread_cond_wait(&R.cv,&R.mu);
      if (!J->done || J->status)
        rc = 1;
      pthread_mutex_unlock(&R.mu);
      w1 = fga_wall();
      twait += w1-w0;
      if (rc == 0 && J->out.n > 0 && fwrite(J->out.s,1,J->out.n,f) != J->out.n)
        { fga_set_error("write error on %s",to_stdout ? "stdout" : path);
          pthread_mutex_lock(&R.mu);
          R.failed = 1;
          pthread_mutex_unlock(&R.mu);
          rc = 1;
        }
      tfile += fga_wall()-w1;
      free(J->out.s);
      J->out.s = NULL;
    }
  for (t = 0; t < nthreads; t++)
    if (tstarted[t]) pthread_join(th[t],NULL);
  if (to_stdout)
    fflush(f);
  else
    { if (fclose(f) != 0 && rc == 0)
        { fga_set_error("write error on %s",path);
          rc = 1;
        }
      if (rc != 0)                                   /* a half-written regular file goes; a FIFO or device the caller named stays */
        { struct stat sb;
          if (stat(path,&sb) == 0 && S_ISREG(sb.st_mode))
            remove(path);
        }
    }
  if (getenv("FGA_PAF_TIMING") != NULL)
    fprintf(stderr,"  write_lines: %d threads, %d chunks: %.1f ms (writer waited %.1f ms for chunks, wrote for %.1f ms)\n",
            nthreads,nchunk,1e3*(fga_wall()-t0),1e3*twait,1e3*tfile);
  for (c = 0; c < nchunk; c++)
    free(R.chunk[c].out.s);
  pthread_mutex_destroy(&R.mu);
  pthread_cond_destroy(&R.cv);
  free(R.chunk); free(th); free(tstarted);
  return rc;
}

int fga_write_paf(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns,
                  const fga_traces *traces, int flags, int nthreads)
{ return write_lines(path,g1,g2,alns,traces,flags,nthreads,0); }

int fga_write_psl(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns,
                  const fga_traces *traces, int nthreads)
{ return write_lines(path,g1,g2,alns,traces,0,nthreads,1); }

/* the regrouping alone, applied in place to a whole set: Path.trace / Path.diffs after Gap_Improver */
int fga_gap_improve(const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns, fga_traces *traces)
{ fmt_scratch W;
  int64_t  i;
  int      rc = 1;

  if (g2 == NULL) g2 = g1;
  if (traces == NULL || traces->naln != alns->naln)
    { fga_set_error("fga_gap_improve: the edit scripts do not belong to this alignment set");
      return 1;
    }
  if (check_records("fga_gap_improve",g1,g2,alns))
    return 1;
  unpack_tables();
  memset(&W,0,sizeof(W));
  for (i = 0; i < alns->naln; i++)
    { const fga_aln *a = alns->alns+i;
      const int comp = (a->flags & 0x1) != 0, n = a->bepos-a->bbpos;
      const uint8_t *A1, *B1;
      int diffs = traces->diffs[i];
      if (todo_from(traces,i) < 0)
        continue;
      A1 = load_a_window(g1,a->aread,a->abpos,a->aepos,traces->tlen[i],&W);
      if (A1 == NULL) goto oom;
      if (n+64 > W.bcap)
        { W.bcap = 2*(int64_t) n + 4096;
          free(W.bbuf);
          W.bbuf = malloc(W.bcap);
          if (W.bbuf == NULL) { W.bcap = 0; goto oom; }
        }
      B1 = load_piece(g2,a->bread,a->bbpos,a->bepos,comp,W.bbuf);
      if (gap_regroup(A1,(int) g1->contigs[a->aread].clen,B1,(int) g2->contigs[a->bread].clen,a->abpos,a->bbpos,
                      traces->trace+traces->toff[i],traces->tlen[i],todo_from(traces,i),&diffs,&W.S))
        goto oom;
      traces->diffs[i] = diffs;
      if (traces->resume != NULL)
        traces->resume[i] = -1;
    }
  rc = 0;
  goto done;
oom:
  fga_set_error("fga_gap_improve: out of memory");
done:
  fmt_free(&W);
  return rc;
}
