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

static int op_add(oplist *L, char op, int len)
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

/* columns k..k+len of A against h..h+len of B as one M, or as =/X runs */
static int op_block(oplist *L, const uint8_t *A1, const uint8_t *B1, int k, int h, int len, int eqx)
{ int i;
  if (!eqx)
    return op_add(L,'M',len);
  for (i = 0; i < len; )
    { const uint8_t *a = A1+k, *b = B1+h;
      int j = i;
      while (j+8 <= len)                          /* equal columns eight at a time */
        { uint64_t x, y;
          memcpy(&x,a+j,8); memcpy(&y,b+j,8);
          if ((x ^= y) != 0)
            { j += __builtin_ctzll(x) >> 3;
              break;
            }
          j += 8;
        }
      while (j < len && a[j] == b[j])
        j += 1;
      if (op_add(L,'=',j-i)) return 1;
      i = j;
      while (j < len && a[j] != b[j])
        j += 1;
      if (op_add(L,'X',j-i)) return 1;
      i = j;
    }
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

typedef struct
  { const fga_gdb *g1, *g2;
    const fga_alns *alns;
    const fga_traces *tr;
    int      flags;
    int      psl;            /* 1: PSL lines (ALNtoPSL.c:77-405) instead of PAF */
    int64_t  beg, end;
    text     out;
    int      status;
  } paf_job;

static void put_divergence(text *X, const fga_aln *a, int64_t iid)
{ const int64_t len = a->aepos-a->abpos;
  const int64_t v = 10000 + (len > 0 ? (10000ll*(len-iid))/len : 0);      /* an empty A interval: dv 0 */
  tx_str(X,"\tdv:f:0.");
  tx_char(X,(char) ('0'+(v/1000)%10));
  tx_char(X,(char) ('0'+(v/100)%10));
  tx_char(X,(char) ('0'+(v/10)%10));
  tx_char(X,(char) ('0'+v%10));
}

static void *paf_thread(void *arg)
{ paf_job *J = arg;
  const fga_gdb *g1 = J->g1, *g2 = J->g2;
  const int cigar_m = (J->flags & FGA_PAF_CIGAR_M) != 0, cigar_x = (J->flags & FGA_PAF_CIGAR_X) != 0;
  const int cs_s = (J->flags & FGA_PAF_CS_SHORT) != 0, cs_l = (J->flags & FGA_PAF_CS_LONG) != 0;
  const int swap = (J->flags & FGA_PAF_SWAP) != 0;
  const int cigar = cigar_m || cigar_x, cs = cs_s || cs_l, bases = cigar || cs;
  static const char lower[4] = { 'a','c','g','t' }, upper[4] = { 'A','C','G','T' };
  oplist   L = { NULL, NULL, 0, 0 };
  scratch  S = { NULL, 0 };
  text    *X = &J->out;
  uint8_t *abuf = NULL, *bbuf = NULL, *arev = NULL;
  int32_t *tcopy = NULL;
  int64_t  tcap = 0, bcap = 0, i;
  int      alast = -1;
  const uint8_t *A1 = NULL;

  if (bases)
    { abuf = malloc(g1->maxctg+4);
      if (abuf == NULL) goto oom;
    }
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
        int diffs = J->tr->diffs[i], del = 0, q, j, from, to, step;
        const uint8_t *B1, *Aw, *Bw;
        int64_t block, iid;

        if (a->aread != alast)
          { A1 = fga_gdb_get_contig(g1,a->aread,abuf) - 1;
            alast = a->aread;
          }
        if (n+4 > bcap)
          { bcap = 2*(int64_t) n + 4096;
            free(bbuf);
            bbuf = malloc(bcap);
            if (bbuf == NULL) goto oom;
          }
        B1 = load_piece(g2,a->bread,a->bbpos,a->bepos,comp,bbuf);
        if (T+1 > tcap)
          { tcap = 2*(int64_t) T + 1024;
            free(tcopy);
            tcopy = malloc(sizeof(int32_t)*tcap);
            if (tcopy == NULL) goto oom;
          }
        memcpy(tcopy,J->tr->trace+J->tr->toff[i],sizeof(int32_t)*T);
        if (todo_from(J->tr,i) >= 0 &&
            gap_regroup(A1,(int) ca->clen,B1,blen,a->abpos,a->bbpos,tcopy,T,todo_from(J->tr,i),&diffs,&S)) goto oom;
        if (build_ops(&L,a,tcopy,T,A1,B1,!(cigar_m && !cs),&del)) goto oom;

        block = (a->aepos-a->abpos) + del;
        iid = block-diffs;
        tx_char(X,'\t'); tx_int(X,iid); tx_char(X,'\t'); tx_int(X,block); tx_str(X,"\t255");
        put_divergence(X,a,iid);
        tx_str(X,"\tdf:i:"); tx_int(X,diffs);

        if (swap)
          for (q = 0; q < L.n; q++)
            L.op[q] = L.op[q] == 'I' ? 'D' : (L.op[q] == 'D' ? 'I' : L.op[q]);
        from = 0; to = L.n; step = 1;
        if (comp && !swap)
          { from = L.n-1; to = -1; step = -1; }

        if (cigar)
          { if (tx_room(X,(size_t) L.n*12+64)) goto oom;
            tx_str(X,"\tcg:Z:");
            if (cigar_m && cs)                         /* =/X runs were built for the cs tag: print them merged */
              { int run = 0;
                for (q = from; q != to; q += step)
                  if (L.op[q] == 'I' || L.op[q] == 'D')
                    { if (run > 0) { tx_int(X,run); tx_char(X,'M'); run = 0; }
                      tx_int(X,L.ln[q]); tx_char(X,L.op[q]);
                    }
                  else
                    run += L.ln[q];
                if (run > 0) { tx_int(X,run); tx_char(X,'M'); }
              }
            else
              for (q = from; q != to; q += step)
                { tx_int(X,L.ln[q]); tx_char(X,L.op[q]); }
          }

        if (cs)
          { const int alen_seg = a->aepos-a->abpos;
            Aw = A1+a->abpos+1;
            Bw = B1+a->bbpos+1;
            if (comp && !swap)                         /* read both segments on the other strand, operations reversed */
              { uint8_t *r;
                free(arev);
                arev = malloc((size_t) alen_seg + n + 8);
                if (arev == NULL) goto oom;
                r = arev;
                for (j = 0; j < alen_seg; j++) r[j] = (uint8_t) (3-Aw[alen_seg-1-j]);
                for (j = 0; j < n; j++) r[alen_seg+j] = (uint8_t) (3-Bw[n-1-j]);
                Aw = r; Bw = r+alen_seg;
              }
            if (swap)
              { const uint8_t *c = Aw; Aw = Bw; Bw = c; }
            if (tx_room(X,(size_t) 3*(alen_seg+n) + (size_t) L.n*12 + 64)) goto oom;
            tx_str(X,"\tcs:Z:");
            for (q = from; q != to; q += step)
              { const int l = L.ln[q];
                switch (L.op[q])
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
  J->status = 0;
  goto done;

oom:
  fga_set_error("fga_write_paf: out of memory");
  J->status = 1;
done:
  free(L.op); free(L.ln); free(S.buf); free(abuf); free(bbuf); free(arev); free(tcopy);
  return NULL;
}


/* One PSL line per alignment (gen_psl, ALNtoPSL.c:77-405): counts, strand, names and ranges, then the ungapped blocks.
 * Always base-level: the edit script is regrouped first, trailing indels at the very end are trimmed off. */
static void *psl_thread(void *arg)
{ paf_job *J = arg;
  const fga_gdb *g1 = J->g1, *g2 = J->g2;
  scratch  S = { NULL, 0 };
  text    *X = &J->out;
  uint8_t *abuf = NULL, *bbuf = NULL;
  int32_t *tcopy = NULL, *blk = NULL;
  int64_t  tcap = 0, bcap = 0, kcap = 0, i;
  int      alast = -1;
  const uint8_t *A1 = NULL;

  abuf = malloc(g1->maxctg+4);
  if (abuf == NULL) goto oom;
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
      const uint8_t *B1;
      int64_t boff;

      if (a.aread != alast)
        { A1 = fga_gdb_get_contig(g1,a.aread,abuf) - 1;
          alast = a.aread;
        }
      if (n+16 > bcap)
        { bcap = 2*(int64_t) n + 4096;
          free(bbuf);
          bbuf = malloc(bcap);
          if (bbuf == NULL) goto oom;
        }
      B1 = load_piece(g2,a.bread,a.bbpos,a.bepos,comp,bbuf);
      if (T+1 > tcap)
        { tcap = 2*(int64_t) T + 1024;
          free(tcopy);
          tcopy = malloc(sizeof(int32_t)*tcap);
          if (tcopy == NULL) goto oom;
        }
      memcpy(tcopy,J->tr->trace+J->tr->toff[i],sizeof(int32_t)*T);
      if (todo_from(J->tr,i) >= 0 &&
          gap_regroup(A1,(int) ca->clen,B1,(int) cb->clen,a.abpos,a.bbpos,tcopy,T,todo_from(J->tr,i),&diffs,&S)) goto oom;

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

      if (T+2 > kcap)
        { kcap = 2*(int64_t) T + 1024;
          free(blk);
          blk = malloc(sizeof(int32_t)*3*kcap);
          if (blk == NULL) goto oom;
        }
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
  J->status = 0;
  goto done;

oom:
  fga_set_error("fga_write_psl: out of memory");
  J->status = 1;
done:
  free(S.buf); free(abuf); free(bbuf); free(tcopy); free(blk);
  return NULL;
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

static int write_lines(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *alns,
                       const fga_traces *traces, int flags, int nthreads, int psl)
{ const int bases = psl || (flags & (FGA_PAF_CIGAR_M|FGA_PAF_CIGAR_X|FGA_PAF_CS_SHORT|FGA_PAF_CS_LONG)) != 0;
  paf_job  *job;
  pthread_t *th;
  FILE     *f;
  int       t, rc = 0;
  int64_t   total = 0, acc = 0, i, nxt;
  double    t0, t1;

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
    { fga_set_error("%s: base-level output needs the edit scripts of fga_trace_pts for the same alignments",
                    psl ? "fga_write_psl" : "fga_write_paf");
      return 1;
    }
  if (check_records(psl ? "fga_write_psl" : "fga_write_paf",g1,g2,alns))
    return 1;
  unpack_tables();
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 64) nthreads = 64;
  if (alns->naln < nthreads) nthreads = alns->naln > 0 ? (int) alns->naln : 1;
  job = calloc(nthreads,sizeof(paf_job));
  th = calloc(nthreads,sizeof(pthread_t));
  if (job == NULL || th == NULL)
    { free(job); free(th);
      fga_set_error("out of memory");
      return 1;
    }
  for (i = 0; i < alns->naln; i++)                   /* equal shares of aligned bases, contiguous in file order */
    total += alns->alns[i].aepos-alns->alns[i].abpos + 200;
  nxt = 0;
  for (t = 0, i = 0; t < nthreads; t++)
    { job[t].g1 = g1; job[t].g2 = g2; job[t].alns = alns; job[t].tr = traces; job[t].flags = flags; job[t].psl = psl;
      job[t].beg = i;
      nxt += total/nthreads + 1;
      while (i < alns->naln && (acc < nxt || t == nthreads-1))
        { acc += alns->alns[i].aepos-alns->alns[i].abpos + 200; i++; }
      job[t].end = i;
    }
  t0 = fga_wall();
  for (t = 1; t < nthreads; t++)
    if (pthread_create(th+t,NULL,psl ? psl_thread : paf_thread,job+t) != 0)
      { (psl ? psl_thread : paf_thread)(job+t); th[t] = 0; }
  (psl ? psl_thread : paf_thread)(job);
  for (t = 1; t < nthreads; t++)
    if (th[t]) pthread_join(th[t],NULL);
  for (t = 0; t < nthreads; t++)
    rc |= job[t].status;
  t1 = fga_wall();
  if (rc == 0)
    { const int to_stdout = (path == NULL || strcmp(path,"-") == 0);
      f = to_stdout ? stdout : fopen(path,"w");
      if (f == NULL)
        { fga_set_error("cannot create %s",path);
          rc = 1;
        }
      else
        { for (t = 0; t < nthreads && rc == 0; t++)
            if (job[t].out.n > 0 && fwrite(job[t].out.s,1,job[t].out.n,f) != job[t].out.n)
              { fga_set_error("write error on %s",to_stdout ? "stdout" : path);
                rc = 1;
              }
          if (to_stdout) fflush(f); else if (fclose(f) != 0 && rc == 0)
            { fga_set_error("write error on %s",path);
              rc = 1;
            }
        }
    }
  if (getenv("FGA_PAF_TIMING") != NULL)
    fprintf(stderr,"  write_lines: %d threads format %.1f ms, file %.1f ms\n",nthreads,1e3*(t1-t0),1e3*(fga_wall()-t1));
  for (t = 0; t < nthreads; t++)
    free(job[t].out.s);
  free(job); free(th);
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
{ scratch  S = { NULL, 0 };
  uint8_t *abuf, *bbuf = NULL;
  int64_t  bcap = 0, i;
  int      alast = -1, rc = 1;
  const uint8_t *A1 = NULL;

  if (g2 == NULL) g2 = g1;
  if (traces == NULL || traces->naln != alns->naln)
    { fga_set_error("fga_gap_improve: the edit scripts do not belong to this alignment set");
      return 1;
    }
  if (check_records("fga_gap_improve",g1,g2,alns))
    return 1;
  unpack_tables();
  abuf = malloc(g1->maxctg+4);
  if (abuf == NULL) goto oom;
  for (i = 0; i < alns->naln; i++)
    { const fga_aln *a = alns->alns+i;
      const int comp = (a->flags & 0x1) != 0, n = a->bepos-a->bbpos;
      const uint8_t *B1;
      int diffs = traces->diffs[i];
      if (a->aread != alast)
        { A1 = fga_gdb_get_contig(g1,a->aread,abuf) - 1;
          alast = a->aread;
        }
      if (n+16 > bcap)
        { bcap = 2*(int64_t) n + 4096;
          free(bbuf);
          bbuf = malloc(bcap);
          if (bbuf == NULL) goto oom;
        }
      B1 = load_piece(g2,a->bread,a->bbpos,a->bepos,comp,bbuf);
      if (todo_from(traces,i) < 0)
        continue;
      if (gap_regroup(A1,(int) g1->contigs[a->aread].clen,B1,(int) g2->contigs[a->bread].clen,a->abpos,a->bbpos,
                      traces->trace+traces->toff[i],traces->tlen[i],todo_from(traces,i),&diffs,&S))
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
  free(S.buf); free(abuf); free(bbuf);
  return rc;
}
