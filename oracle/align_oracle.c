/* align_oracle.c -- TEST INFRASTRUCTURE (oracle/): CPU restatement of FastGA's wave-based local alignment.
 *
 * Restates Local_Alignment / forward_wave / reverse_wave / New_Align_Spec / set_table of the reference
 * (align.c:1423-1576, 352-874, 878-1418, 222-268, 207-218) in the *lane-parallel* form the HIP kernel uses:
 * every wave step first computes, for all diagonals independently, the new furthest-reaching point from the
 * PREVIOUS wave's values (predecessor choice, 60-column match history, snake, trace-point pebbles), and only
 * then runs the ordered "new best point" scan, the sequence-end clipping and the WAVE_LAG pruning.  Forward
 * and reverse waves share one routine parameterised by the direction s = +1 / -1 (the reverse wave is the
 * exact mirror: minimise instead of maximise, sweep k upwards, compare A[x-1]/B[x-1-k], trace points walk
 * down).  It is checked call-by-call against the real reference (oracle/_ref/libalign_ref.so) by
 * tests/test_oracle_vs_reference.py.
 *
 * Nothing under fastga_amd/ may call this; it is the checker for tests/, smoke() and bench.py's cpu_baseline.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TRIM_LEN    15
#define DUB_TRIM    45
#define PATH_LEN    60
#define PATH_TOP    0x1000000000000000ull
#define PATH_INT    0x0fffffffffffffffull
#define TRIM_MASK   0x7fff
#define TRIM_MLAG   250
#define WAVE_LAG    70
#define FRACTION    1000
#define BIG         0x7fffffff

/* ---------------------------------------------------------------------------------------------------
 *  Alignment spec: PATH_AVE and the two 32K-entry int16 tables (align.c:178-268)
 * --------------------------------------------------------------------------------------------------- */

typedef struct
  { int      ave_path;
    int      tspace;
    int      reach;
    int16_t  score[TRIM_MASK+1];
    int16_t  table[TRIM_MASK+1];
  } oracle_spec;

static void fill_table(int bit, int prefix, int score, int max, int mscore, int dscore, oracle_spec *sp)
{ if (bit >= TRIM_LEN)
    { sp->table[prefix] = (int16_t) (score-max);
      sp->score[prefix] = (int16_t) score;
    }
  else
    { if (score > max)
        max = score;
      fill_table(bit+1,(prefix<<1),score - dscore,max,mscore,dscore,sp);
      fill_table(bit+1,(prefix<<1) | 1,score + mscore,max,mscore,dscore,sp);
    }
}

void oracle_align_spec(double ave_corr, int tspace, const float *freq, int reach, oracle_spec *sp)
{ static const double Bias_Factor[10] = { .690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000 };
  double match;
  int    bias, mscore, dscore;

  match = freq[0] + freq[3];
  if ((match <= 0.) == (match > 0.))
    match = .5;
  if (match > .5)
    match = 1.-match;
  bias = (int) ((match+.025)*20.-1.);
  if (match < .2)
    bias = 3;
  sp->tspace   = tspace;
  sp->reach    = reach;
  sp->ave_path = (int) (PATH_LEN * (1. - Bias_Factor[bias] * (1. - ave_corr)));
  mscore = (int) (FRACTION * Bias_Factor[bias] * (1. - ave_corr));
  dscore = FRACTION - mscore;
  fill_table(0,0,0,0,mscore,dscore,sp);
}

/* ---------------------------------------------------------------------------------------------------
 *  Wave state
 * --------------------------------------------------------------------------------------------------- */

typedef struct { int ptr, diag, diff, mark; } pebble;

typedef struct
  { int       cap;          /* ring size (power of two) for the per-diagonal arrays */
    int      *V[2];
    int      *M[2];
    int      *HA[2];
    uint64_t *T[2];
    int      *NA;
    pebble   *cells;
    int       cmax, avail;
  } wave_work;

static int work_init(wave_work *w)
{ int i;
  memset(w,0,sizeof(*w));
  w->cap = 4096;
  for (i = 0; i < 2; i++)
    { w->V[i]  = malloc(sizeof(int)*w->cap);
      w->M[i]  = malloc(sizeof(int)*w->cap);
      w->HA[i] = malloc(sizeof(int)*w->cap);
      w->T[i]  = malloc(sizeof(uint64_t)*w->cap);
    }
  w->NA = malloc(sizeof(int)*w->cap);
  w->cmax = 1<<16;
  w->cells = malloc(sizeof(pebble)*w->cmax);
  return w->cells == NULL;
}

static void work_free(wave_work *w)
{ int i;
  for (i = 0; i < 2; i++)
    { free(w->V[i]); free(w->M[i]); free(w->HA[i]); free(w->T[i]); }
  free(w->NA); free(w->cells);
}

static int new_pebble(wave_work *w, int ptr, int diag, int diff, int mark)
{ if (w->avail >= w->cmax)
    { w->cmax = (int) (w->cmax*1.5) + 10000;
      w->cells = realloc(w->cells,sizeof(pebble)*w->cmax);
      if (w->cells == NULL)
        return -2;
    }
  w->cells[w->avail].ptr  = ptr;
  w->cells[w->avail].diag = diag;
  w->cells[w->avail].diff = diff;
  w->cells[w->avail].mark = mark;
  return w->avail++;
}

typedef struct
  { const uint8_t *aseq, *bseq;    /* numeric 0..3, value 4 at index -1 and at index len */
    int alen, blen;
    int abpos, bbpos, aepos, bepos, diffs, tlen;
    uint16_t *trace;               /* points into the middle of a caller buffer */
  } walign;

#define IX(k) ((k) & msk)

/* One directional wave extension from the points on anti-diagonal `mida`, diagonals [*mind,maxd].
 * s = +1: forward_wave (align.c:352-874);  s = -1: reverse_wave (align.c:878-1418).               */
static int wave(wave_work *W, const oracle_spec *sp, walign *al, int s,
                int *mind, int maxd, int mida, int minp, int maxp, int aoff)
{ const uint8_t *aseq = al->aseq, *bseq = al->bseq;
  const int tspace = sp->tspace, PATH_AVE = sp->ave_path;
  const int16_t *SCORE = sp->score, *TABLE = sp->table;
  const int msk = W->cap-1;
  const int VNEW = (s > 0) ? -1 : BIG;

  int low = *mind, hgh = maxd, dif = 0, cur = 0;
  int more = 1;
  int aclip, bclip;
  int besta, bestx, trima, trimx, trimd, trimha, morea, morex, mored, moreha, morem, lasta;
  int k;

  W->avail = 0;
  aclip = (s > 0) ?  BIG : -BIG;
  bclip = (s > 0) ? -BIG :  BIG;

  besta = trima = morea = lasta = mida;
  bestx = trimx = morex = (mida+hgh)>>1;
  trimd = mored = 0;
  trimha = moreha = 0;
  morem = -1;

#define SEQ_STEP(x,k,hitA,hitB,stop)                                                      \
  { int _c, _d;                                                                           \
    if (s > 0) { _c = bseq[(x)-(k)]; _d = aseq[(x)]; }                                    \
    else       { _c = bseq[(x)-1-(k)]; _d = aseq[(x)-1]; }                                \
    hitA = hitB = 0; stop = 0;                                                            \
    if (_c == 4) { hitB = 1; stop = 1; }                                                  \
    else if (_c != _d) { if (_d == 4) hitA = 1; stop = 1; }                               \
  }

  /* wave 0 (align.c:425-512 / 949-1035): order of k is irrelevant except for the strict-best scan */
  { int *V = W->V[cur], *M = W->M[cur], *HA = W->HA[cur];
    uint64_t *T = W->T[cur];
    int j, span = hgh-low+1;
    for (j = 0; j < span; j++)
      { int x, c, ha, na, hA, hB, stop;
        k = (s > 0) ? hgh-j : low+j;
        x = (mida+k)>>1;
        if (s > 0)
          { na = ((x+(tspace-aoff))/tspace-1)*tspace+aoff;
            ha = new_pebble(W,-1,k,0,na);
            na += tspace;
          }
        else
          { na = ((x+(tspace-aoff)-1)/tspace-1)*tspace+aoff;
            ha = new_pebble(W,-1,k,0,x);
          }
        if (ha < -1) return 1;
        while (1)
          { SEQ_STEP(x,k,hA,hB,stop);
            if (stop)
              { if (hB) { more = 0; if (s > 0 ? bclip < k : bclip > k) bclip = k; }
                if (hA) { more = 0; aclip = k; }
                break;
              }
            x += s;
          }
        c = (x << 1) - k;
        while (s > 0 ? x >= na : x <= na)
          { ha = new_pebble(W,ha,k,0,na);
            if (ha < -1) return 1;
            na += s*tspace;
          }
        if (s > 0 ? c > besta : c < besta)
          { besta = trima = lasta = c;
            bestx = trimx = x;
            trimha = ha;
          }
        V[IX(k)] = c; T[IX(k)] = PATH_INT; M[IX(k)] = PATH_LEN; HA[IX(k)] = ha; W->NA[IX(k)] = na;
      }
  }

#define CLIP_UPDATE(withd)                                                                 \
  if (more == 0)                                                                           \
    { int *V = W->V[cur], *M = W->M[cur], *HA = W->HA[cur];                                \
      uint8_t cb = (s > 0) ? bseq[besta-bestx] : bseq[besta-bestx-1];                      \
      uint8_t ca = (s > 0) ? aseq[bestx] : aseq[bestx-1];                                  \
      if (cb != 4 && ca != 4)                                                              \
        more = 1;                                                                          \
      if (s > 0)                                                                           \
        { if (hgh >= aclip)                                                                \
            { hgh = aclip-1;                                                               \
              if (morem <= M[IX(aclip)])                                                   \
                { morem = M[IX(aclip)]; morea = V[IX(aclip)]; morex = (morea+aclip)>>1;    \
                  if (withd) mored = dif;                                                  \
                  moreha = HA[IX(aclip)];                                                  \
                }                                                                          \
            }                                                                              \
          if (low <= bclip)                                                                \
            { low = bclip+1;                                                               \
              if (morem <= M[IX(bclip)])                                                   \
                { morem = M[IX(bclip)]; morea = V[IX(bclip)]; morex = (morea+bclip)>>1;    \
                  if (withd) mored = dif;                                                  \
                  moreha = HA[IX(bclip)];                                                  \
                }                                                                          \
            }                                                                              \
          aclip = BIG; bclip = -BIG;                                                       \
        }                                                                                  \
      else                                                                                 \
        { if (low <= aclip)                                                                \
            { low = aclip+1;                                                               \
              if (morem <= M[IX(aclip)])                                                   \
                { morem = M[IX(aclip)]; morea = V[IX(aclip)]; morex = (morea+aclip)>>1;    \
                  if (withd) mored = dif;                                                  \
                  moreha = HA[IX(aclip)];                                                  \
                }                                                                          \
            }                                                                              \
          if (hgh >= bclip)                                                                \
            { hgh = bclip-1;                                                               \
              if (morem <= M[IX(bclip)])                                                   \
                { morem = M[IX(bclip)]; morea = V[IX(bclip)]; morex = (morea+bclip)>>1;    \
                  if (withd) mored = dif;                                                  \
                  moreha = HA[IX(bclip)];                                                  \
                }                                                                          \
            }                                                                              \
          aclip = -BIG; bclip = BIG;                                                       \
        }                                                                                  \
    }

  CLIP_UPDATE(0)

  /* successive waves (align.c:546-803 / 1067-1323) */
  while (more && (s > 0 ? lasta >= besta - TRIM_MLAG : lasta <= besta + TRIM_MLAG))
    { int *Vo, *Mo, *HAo, *Vn, *Mn, *HAn;
      uint64_t *To, *Tn;
      int j, span;

      if (hgh-low+8 >= W->cap)
        return 2;                           /* ring too small: not expected (width is bounded by pruning) */

      low -= 1;
      hgh += 1;
      Vo = W->V[cur]; Mo = W->M[cur]; HAo = W->HA[cur]; To = W->T[cur];
      if (low >= minp)
        { W->NA[IX(low)] = W->NA[IX(low+1)]; Vo[IX(low)] = VNEW; }
      else
        low += 1;
      if (hgh <= maxp)
        { W->NA[IX(hgh)] = W->NA[IX(hgh-1)]; Vo[IX(hgh)] = VNEW; }
      else
        hgh -= 1;
      dif += 1;
      Vo[IX(hgh+1)] = Vo[IX(low-1)] = VNEW;

      Vn = W->V[cur^1]; Mn = W->M[cur^1]; HAn = W->HA[cur^1]; Tn = W->T[cur^1];
      span = hgh-low+1;

      /* (1) every diagonal independently, from the previous wave's values */
      for (j = 0; j < span; j++)
        { int ac, a1, a2, c, m, ha, x, src;
          uint64_t b;
          k  = (s > 0) ? hgh-j : low+j;
          ac = Vo[IX(k)];
          a1 = Vo[IX(k-s)];          /* second priority: V[k-1] forward, V[k+1] reverse */
          a2 = Vo[IX(k+s)];          /* third priority */
          if (s > 0)
            { if (ac < a1) src = (a1 < a2) ? k+s : k-s;
              else         src = (ac < a2) ? k+s : k;
            }
          else
            { if (ac > a1) src = (a1 > a2) ? k+s : k-s;
              else         src = (ac > a2) ? k+s : k;
            }
          if (src == k) c = ac + 2*s; else c = Vo[IX(src)] + s;
          if (src == k+s && ((s > 0) ? (k+s > hgh) : (k+s < low)))
            { m = PATH_LEN; b = PATH_INT; ha = -1; }       /* (n,t,ua) initial values, align.c:636-638 */
          else
            { m = Mo[IX(src)]; b = To[IX(src)]; ha = HAo[IX(src)]; }

          if ((b & PATH_TOP) != 0)
            m -= 1;
          b <<= 1;

          x = (c+k)>>1;
          while (1)
            { int hA, hB, stop;
              SEQ_STEP(x,k,hA,hB,stop);
              if (stop)
                { if (hB) { more = 0; if (s > 0 ? bclip < k : bclip > k) bclip = k; }
                  if (hA) { more = 0; aclip = k; }
                  break;
                }
              x += s;
              if ((b & PATH_TOP) == 0)
                m += 1;
              b = (b << 1) | 1;
            }
          c = (x << 1) - k;

          while (s > 0 ? x >= W->NA[IX(k)] : x <= W->NA[IX(k)])
            { int mk = W->cells[ha].mark;
              if (s > 0 ? mk < W->NA[IX(k)] : mk > W->NA[IX(k)])
                { ha = new_pebble(W,ha,k,dif,W->NA[IX(k)]);
                  if (ha < -1) return 1;
                }
              W->NA[IX(k)] += s*tspace;
            }
          Vn[IX(k)] = c; Tn[IX(k)] = b; Mn[IX(k)] = m; HAn[IX(k)] = ha;
        }

      /* (2) ordered scan: strict new best points, in sweep order (align.c:729-742) */
      for (j = 0; j < span; j++)
        { int c, x;
          uint64_t b;
          k = (s > 0) ? hgh-j : low+j;
          c = Vn[IX(k)];
          if (s > 0 ? c > besta : c < besta)
            { x = (c+k)>>1;
              b = Tn[IX(k)];
              besta = c;
              bestx = x;
              if (Mn[IX(k)] >= PATH_AVE)
                { lasta = c;
                  if (TABLE[b & TRIM_MASK] >= 0)
                    if (TABLE[(b >> TRIM_LEN) & TRIM_MASK] + SCORE[b & TRIM_MASK] >= 0)
                      { trima = c; trimx = x; trimd = dif; trimha = HAn[IX(k)]; }
                }
            }
        }
      cur ^= 1;

      /* aclip: the LAST diagonal in sweep order that hit an A end; handled above by plain assignment
         in sweep order -- the loop (1) runs in sweep order here, so this is identical.               */
      CLIP_UPDATE(1)

      /* (3) prune both ends (align.c:782-790) */
      { int *V = W->V[cur];
        int n = besta - s*WAVE_LAG;
        while (hgh >= low)
          if (s > 0 ? V[IX(hgh)] < n : V[IX(hgh)] > n)
            hgh -= 1;
          else
            { while (s > 0 ? V[IX(low)] < n : V[IX(low)] > n)
                low += 1;
              break;
            }
      }
    }

  /* unwind the pebble list into trace pairs (align.c:805-870 / 1325-1415) */
  { uint16_t *atrace = al->trace;
    pebble *cells = W->cells;
    int atlen, trimy, a, b, h, d, e;

    if (morem >= 0 && sp->reach)
      { trimx = morex; trimy = morea - morex; trimd = mored; trimha = moreha; }
    else
      trimy = trima - trimx;

    atlen = 0;
    a = -1;
    for (h = trimha; h >= 0; h = b)
      { b = cells[h].ptr;
        cells[h].ptr = a;
        a = h;
      }
    h = a;
    k = cells[h].diag;

    if (s > 0)
      { b = (mida-k)>>1;
        e = 0;
        low = k;
        for (h = cells[h].ptr; h >= 0; h = cells[h].ptr)
          { k = cells[h].diag;
            a = cells[h].mark - k;
            d = cells[h].diff;
            atrace[atlen++] = (uint16_t) (d-e);
            atrace[atlen++] = (uint16_t) (a-b);
            b = a;
            e = d;
          }
        if (b+k != trimx)
          { atrace[atlen++] = (uint16_t) (trimd-e);
            atrace[atlen++] = (uint16_t) (trimy-b);
          }
        else if (b != trimy)
          { atrace[atlen-1] = (uint16_t) (atrace[atlen-1] + (trimy-b));
            atrace[atlen-2] = (uint16_t) (atrace[atlen-2] + (trimd-e));
          }
        al->aepos = trimx;
        al->bepos = trimy;
        al->diffs = trimd;
        al->tlen  = atlen;
        *mind = low;
      }
    else
      { b = cells[h].mark - k;
        e = 0;
        if ((b+k)%tspace != aoff)
          { h = cells[h].ptr;
            if (h < 0)
              { a = trimy; d = trimd; }
            else
              { k = cells[h].diag;
                a = cells[h].mark - k;
                d = cells[h].diff;
              }
            if (al->tlen == 0)
              { atrace[--atlen] = (uint16_t) (b-a);
                atrace[--atlen] = (uint16_t) (d-e);
              }
            else
              { atrace[1] = (uint16_t) (atrace[1] + (b-a));
                atrace[0] = (uint16_t) (atrace[0] + (d-e));
              }
            b = a;
            e = d;
          }
        if (h >= 0)
          { for (h = cells[h].ptr; h >= 0; h = cells[h].ptr)
              { k = cells[h].diag;
                a = cells[h].mark - k;
                atrace[--atlen] = (uint16_t) (b-a);
                d = cells[h].diff;
                atrace[--atlen] = (uint16_t) (d-e);
                b = a;
                e = d;
              }
            if (b+k != trimx)
              { atrace[--atlen] = (uint16_t) (b-trimy);
                atrace[--atlen] = (uint16_t) (trimd-e);
              }
            else if (b != trimy)
              { atrace[atlen+1] = (uint16_t) (atrace[atlen+1] + (b-trimy));
                atrace[atlen]   = (uint16_t) (atrace[atlen]   + (trimd-e));
              }
          }
        al->abpos = trimx;
        al->bbpos = trimy;
        al->diffs = al->diffs + trimd;
        al->tlen  = al->tlen - atlen;
        al->trace = atrace + atlen;
      }
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------
 *  Local_Alignment (align.c:1423-1576)
 * --------------------------------------------------------------------------------------------------- */

typedef struct
  { int abpos, bbpos, aepos, bepos, diffs, tlen; } oracle_path;

/* aseq/bseq: numeric with the sentinel 4 at [-1] and [len] (A already complemented for the C pass, in
 * which case acomp != 0).  selfie: the reference's `align->aseq == align->bseq` test.  trace_out must
 * hold 4*(alen/tspace+2)+8 uint16.  Returns 0, 1 (memory) or 2 (internal ring overflow).            */
int oracle_local_alignment(const uint8_t *aseq, int alen, const uint8_t *bseq, int blen,
                           int acomp, int selfie, const oracle_spec *sp,
                           int low, int hgh, int anti, int lbord, int hbord,
                           oracle_path *out, uint16_t *trace_out)
{ wave_work W;
  walign al;
  int maxtp, minp, maxp, aoff, fshort, rshort, st;
  uint16_t *points;

  if (work_init(&W))
    return 1;
  maxtp  = 2*(alen/sp->tspace+2);
  points = calloc((size_t) 4*maxtp+64,sizeof(uint16_t));
  if (points == NULL)
    { work_free(&W);
      return 1;
    }
  memset(&al,0,sizeof(al));
  al.aseq = aseq; al.bseq = bseq; al.alen = alen; al.blen = blen;
  al.trace = points + 2*maxtp + 32;

  while (((anti-hgh)>>1) < 0)
    hgh -= 1;

  if (lbord < 0)
    minp = (selfie && low >= 0) ? 1 : -BIG;
  else
    minp = low-lbord;
  if (hbord < 0)
    maxp = (selfie && hgh <= 0) ? -1 : BIG;
  else
    maxp = hgh+hbord;

  aoff = acomp ? alen % sp->tspace : 0;

  if ((st = wave(&W,sp,&al,+1,&low,hgh,anti,minp,maxp,aoff)) != 0) goto done;
  fshort = ((al.aepos + al.bepos) - anti < DUB_TRIM);
  { int l2 = low;
    if ((st = wave(&W,sp,&al,-1,&l2,low,anti,minp,maxp,aoff)) != 0) goto done;
  }
  rshort = (anti - (al.abpos + al.bbpos) < DUB_TRIM);

  if (fshort)
    { if (rshort)
        { al.aepos = al.abpos = (al.abpos+al.aepos)>>1;
          al.bepos = al.bbpos = (al.bbpos+al.bepos)>>1;
          al.tlen  = 0;
        }
      else
        { low  = al.abpos - al.bbpos;
          anti = al.abpos + al.bbpos;
          al.tlen = 0;
          if ((st = wave(&W,sp,&al,+1,&low,low,anti,minp,maxp,aoff)) != 0) goto done;
        }
    }
  else if (rshort)
    { low  = al.aepos - al.bepos;
      anti = al.aepos + al.bepos;
      al.tlen = 0;
      al.diffs = 0;
      if ((st = wave(&W,sp,&al,-1,&low,low,anti,minp,maxp,aoff)) != 0) goto done;
    }

  if (acomp)
    { uint16_t *trace = al.trace, p;
      int i, j;
      i = al.abpos; al.abpos = alen - al.aepos; al.aepos = alen - i;
      i = al.bbpos; al.bbpos = blen - al.bepos; al.bepos = blen - i;
      i = al.tlen-2;
      j = 0;
      while (j < i)
        { p = trace[i];   trace[i]   = trace[j];   trace[j]   = p;
          p = trace[i+1]; trace[i+1] = trace[j+1]; trace[j+1] = p;
          i -= 2;
          j += 2;
        }
    }
  out->abpos = al.abpos; out->bbpos = al.bbpos; out->aepos = al.aepos; out->bepos = al.bepos;
  out->diffs = al.diffs; out->tlen = al.tlen;
  if (al.tlen > 0)
    memcpy(trace_out,al.trace,sizeof(uint16_t)*al.tlen);
  st = 0;

done:
  free(points);
  work_free(&W);
  return st;
}
