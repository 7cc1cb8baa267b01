/* seed_merge_oracle.c -- TEST INFRASTRUCTURE (oracle/): CPU restatement of FastGA's adaptive seed merge.
 *
 * Restates, as a pure function of (T1 entry, T2 panel), what the reference computes with its streaming
 * state machine:
 *   pair mode  : new_merge_thread       reference FastGA.c:610-1025   (flip = 0 and the -S flip = 1 pass)
 *   self mode  : new_self_merge_thread  reference FastGA.c:1616-1909
 * Semantics (checked against seed streams captured from the real reference by
 * tests/test_oracle_vs_reference.py):
 *   For every entry s of table 1 whose 12-mer panel is non-empty in table 2 (panel C, sorted):
 *     plen = min(40, max_j LCP(s,c_j));  R = { j : LCP(s,c_j) >= plen }  (a contiguous run);
 *     |R| >= FREQ -> nothing (FastGA.c:796-823);   s.mask >= mlen -> nothing (824-832), mlen = plen when
 *     soft masking else 41;
 *     flip = 0: s on the complement strand -> nothing (921-928); else one seed per c in R with
 *               c.mask < mlen: {plen, s.payload, c.payload}, stream C iff c is complement (933-984);
 *     flip = 1: one seed per forward-strand c in R with c.mask < mlen: {plen, c.payload, s.payload},
 *               stream C iff s is complement (833-892).
 *   Self mode: plen = max(L[k],L[k+1]) over the stored LCP bytes (sentinel 11 after the panel), R the
 *     maximal run around k with LCP >= plen (it contains k), same frequency / mask tests; one seed per
 *     c in R other than s itself: {plen, s.payload with the sign bit cleared, c.payload}, stream N iff
 *     the two signs are equal (1801-1864).
 * Seed record bytes: u8 plen; A payload (IBYTE); B payload (JBYTE)  (FastGA.c:961-966).
 *
 * Nothing under fastga_amd/ may call this; it is the checker for tests/, smoke() and bench.py's
 * cpu_baseline leg only.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct
  { uint8_t *buf;
    int64_t  len, cap;
  } stream;

static int push(stream *s, const uint8_t *rec, int n)
{ if (s->len + n > s->cap)
    { s->cap = s->cap*2 + (1<<20);
      s->buf = realloc(s->buf,s->cap);
      if (s->buf == NULL)
        return 1;
    }
  memcpy(s->buf+s->len,rec,n);
  s->len += n;
  return 0;
}

static inline uint64_t suffix56(const uint8_t *e)
{ return ((uint64_t) e[0] << 48) | ((uint64_t) e[1] << 40) | ((uint64_t) e[2] << 32)
       | ((uint64_t) e[3] << 24) | ((uint64_t) e[4] << 16) | ((uint64_t) e[5] << 8) | e[6];
}

static inline int lcp_suf(uint64_t a, uint64_t b)     /* LCP of two k-mers of the same panel, in bases */
{ if (a == b)
    return 40;
  return 12 + ((__builtin_clzll(a^b) - 8) >> 1);
}

/* results are handed back through these; caller frees with oracle_free */
typedef struct
  { uint8_t *nbuf; int64_t nlen;
    uint8_t *cbuf; int64_t clen;
    int64_t  nhits, tseed;
  } oracle_seeds;

void oracle_free(oracle_seeds *o)
{ free(o->nbuf); free(o->cbuf);
  o->nbuf = o->cbuf = NULL;
}

/* tab: nents*E raw on-disk entries; idx: 2^24 inclusive cumulative counts; pbyte = post+contig bytes. */
int oracle_seed_merge(const uint8_t *tab1, const int64_t *idx1, int pbyte1,
                      const uint8_t *tab2, const int64_t *idx2, int pbyte2,
                      int freq, int soft_mask, int flip, int64_t pfirst, int64_t plast,
                      oracle_seeds *out)
{ const int E1 = 9+pbyte1, E2 = 9+pbyte2;
  stream N = { NULL, 0, 0 }, C = { NULL, 0, 0 };
  int64_t p, nhits = 0, tseed = 0;
  uint8_t rec[64];

  for (p = pfirst; p < plast; p++)
    { int64_t a0 = p ? idx1[p-1] : 0, a1 = idx1[p];
      int64_t b0 = p ? idx2[p-1] : 0, b1 = idx2[p];
      int64_t i;
      if (a0 == a1 || b0 == b1)
        continue;
      for (i = a0; i < a1; i++)
        { const uint8_t *s = tab1 + i*E1;
          uint64_t ks = suffix56(s);
          int64_t lo = b0, hi = b1, lb, low, hgh, j;
          int plen, mlen, la, lc;

          while (lo < hi)                        /* lower bound of s among the panel's suffixes */
            { int64_t m = (lo+hi) >> 1;
              if (suffix56(tab2 + m*E2) < ks) lo = m+1; else hi = m;
            }
          lb = lo;
          la = (lb > b0) ? lcp_suf(ks,suffix56(tab2 + (lb-1)*E2)) : 0;
          lc = (lb < b1) ? lcp_suf(ks,suffix56(tab2 + lb*E2)) : 0;
          plen = la > lc ? la : lc;

          low = lb;
          while (low > b0 && lcp_suf(ks,suffix56(tab2 + (low-1)*E2)) >= plen && lb-low <= freq)
            low -= 1;
          hgh = lb;
          while (hgh < b1 && lcp_suf(ks,suffix56(tab2 + hgh*E2)) >= plen && hgh-low <= freq)
            hgh += 1;
          if (hgh-low >= freq)
            continue;
          mlen = soft_mask ? plen : 41;
          if (s[7] >= mlen)
            continue;
          if (!flip)
            { const uint8_t *pay1 = s+9;
              if (pay1[pbyte1-1] & 0x80)
                continue;
              for (j = low; j < hgh; j++)
                { const uint8_t *c = tab2 + j*E2;
                  if (c[7] >= mlen)
                    continue;
                  rec[0] = (uint8_t) plen;
                  memcpy(rec+1,pay1,pbyte1);
                  memcpy(rec+1+pbyte1,c+9,pbyte2);
                  if (push((c[9+pbyte2-1] & 0x80) ? &C : &N,rec,1+pbyte1+pbyte2))
                    return 1;
                  nhits += 1;
                  tseed += plen;
                }
            }
          else                                   /* table 1 is genome 2 here: A side comes from table 2 */
            { const uint8_t *pay1 = s+9;
              int bsign = pay1[pbyte1-1] & 0x80;
              for (j = low; j < hgh; j++)
                { const uint8_t *c = tab2 + j*E2;
                  if ((c[9+pbyte2-1] & 0x80) || c[7] >= mlen)
                    continue;
                  rec[0] = (uint8_t) plen;
                  memcpy(rec+1,c+9,pbyte2);
                  memcpy(rec+1+pbyte2,pay1,pbyte1);
                  if (push(bsign ? &C : &N,rec,1+pbyte1+pbyte2))
                    return 1;
                  nhits += 1;
                  tseed += plen;
                }
            }
        }
    }
  out->nbuf = N.buf; out->nlen = N.len;
  out->cbuf = C.buf; out->clen = C.len;
  out->nhits = nhits; out->tseed = tseed;
  return 0;
}

int oracle_self_seed_merge(const uint8_t *tab, const int64_t *idx, int pbyte,
                           int freq, int soft_mask, int64_t pfirst, int64_t plast, oracle_seeds *out)
{ const int E = 9+pbyte;
  stream N = { NULL, 0, 0 }, C = { NULL, 0, 0 };
  int64_t p, nhits = 0, tseed = 0;
  uint8_t rec[64];

  for (p = pfirst; p < plast; p++)
    { int64_t a0 = p ? idx[p-1] : 0, a1 = idx[p];
      int64_t k;
      for (k = a0; k < a1; k++)
        { const uint8_t *s = tab + k*E;
          int lk  = (k > a0) ? s[8] : 0;
          int lk1 = (k+1 < a1) ? s[E+8] : 11;
          int plen = lk > lk1 ? lk : lk1;
          int64_t low = k, hgh = k+1, j;
          int mlen, isign;

          while (low > a0 && tab[low*E+8] >= plen && k-low <= freq)
            low -= 1;
          while (hgh < a1 && tab[hgh*E+8] >= plen && hgh-low <= freq)
            hgh += 1;
          if (hgh-low >= freq)
            continue;
          mlen = soft_mask ? plen : 41;
          if (s[7] >= mlen)
            continue;
          isign = s[9+pbyte-1] & 0x80;
          for (j = low; j < hgh; j++)
            { const uint8_t *c = tab + j*E;
              int jsign;
              if (j == k || c[7] >= mlen)
                continue;
              jsign = c[9+pbyte-1] & 0x80;
              rec[0] = (uint8_t) plen;
              memcpy(rec+1,s+9,pbyte);
              rec[1+pbyte-1] &= 0x7f;
              memcpy(rec+1+pbyte,c+9,pbyte);
              if (push(isign == jsign ? &N : &C,rec,1+2*pbyte))
                return 1;
              nhits += 1;
              tseed += plen;
            }
        }
    }
  out->nbuf = N.buf; out->nlen = N.len;
  out->cbuf = C.buf; out->clen = C.len;
  out->nhits = nhits/2; out->tseed = tseed/2;
  return 0;
}
