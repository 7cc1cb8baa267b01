/* fga_one.c -- reading binary ONEcode files: line iterator with list codecs, and the .1aln reader built on it.
 *
 * The container is ONElib's (reference ONElib.c): ASCII header (type line, provenance '!', references '<', schema '~')
 * closed by "$ <endian>"; then binary lines: one type byte 0x80 | k<<1 | c (k: A-Z 0-25, a-z 26-51, ';' 52, '&' 53,
 * '/' 54, '.' 55; c: the list is codec-compressed), the fields in schema order (INT = packed integer, REAL = 8 bytes,
 * CHAR = 1 byte, a list = packed length then the elements); an empty line "\n" ends the data, the footer follows
 * (count lines in ASCII, object index '&' and list codes ';' in binary, "^"), and the last 8 bytes hold the footer's
 * offset.  INT_LIST elements: the first as a packed integer, then one width byte w, then w-byte little-endian signed
 * differences (ONElib.c:902-1000).  Compressed lists: packed bit count, then the code stream (see codec_decode).
 *
 * fga_read_1aln (replaces open_Aln_Read + Read_Aln_Overlap + Read_Aln_Trace, alncode.c:62-237, for whole files) turns
 * the A / R / D / T / X lines of a .1aln -- the reference's own, compressed or not, or ours -- into an fga_alns set,
 * which fga_trace_pts / fga_write_paf / fga_write_psl take: together the in-process ALNtoPAF / ALNtoPSL.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "fga_host.h"
#include "fastga_amd.h"

enum { F_INT = 1, F_REAL, F_CHAR, F_STRING, F_INT_LIST, F_REAL_LIST, F_STRING_LIST, F_DNA };

/* ---- packed integers (ONElib.c:3737-3790) ---- */
int fga_one_int(const uint8_t *u, const uint8_t *end, int64_t *val)   /* bytes used, 0 on overrun */
{ int n;
  uint64_t v = 0;
  if (u >= end) return 0;
  switch (u[0] >> 5)
  { case 2: case 3: *val = u[0] & 0x3f; return 1;
    case 6: case 7: *val = (int64_t) (int8_t) u[0]; return 1;
    case 1: if (u+1 >= end) return 0;
            *val = ((int64_t) (u[0] & 0x1f) << 8) | u[1]; return 2;
    case 0: case 4:
      n = (u[0] & 7) + 1;
      if (n < 2 || u+n >= end) return 0;
      memcpy(&v,u+1,(size_t) n);
      if ((u[0] >> 5) == 4 && n < 8)
        v |= ~(uint64_t) 0 << (8*n);
      *val = (int64_t) v;
      return n+1;
    default: return 0;
  }
}

/* ---- list codes (ONElib.c:3305-3420 serialised form, 3621-3725 stream) ----
 * A stream is a sequence of 64-bit words, most significant bit first, each stored in the writer's byte order, followed
 * by the last < 64 bits byte by byte; bytes 0 and 7 of the first word are exchanged on a little-endian writer so that the
 * two flag bits (0x80: always, 0x40: big-endian writer) sit in byte 0; a first byte 0xff means "stored as is".  Symbols
 * are found through their 16-bit prefix; the escape code is followed by a literal byte. */
int fga_one_codec_parse(fga_one_codec *c, const uint8_t *in, int64_t n)
{ const uint8_t *q, *end = in+n;
  uint16_t bits[256];
  int i;
  if (n < 9 || in[0] != 0)                  /* [0] = 1: written on a big-endian machine -- not supported */
    return 1;
  memcpy(&c->esc,in+1,4);
  memcpy(&c->esclen,in+5,4);
  if (c->esc < -1 || c->esc > 255 || c->esclen < 0 || c->esclen > 16)      /* a code word is at most 16 bits */
    return 1;
  q = in+9;
  for (i = 0; i < 256; i++)
    { if (q >= end) return 1;
      c->len[i] = *q++;
      if (c->len[i] > 16) return 1;
      bits[i] = 0;
      if (c->len[i] > 0 || i == c->esc)
        { if (q+2 > end) return 1;
          memcpy(bits+i,q,2);
          q += 2;
        }
    }
  free(c->look);
  c->look = malloc(0x10000);
  if (c->look == NULL) return 1;
  memset(c->look,0,0x10000);
  for (i = 0; i < 256; i++)
    { const int l = (i == c->esc) ? c->esclen : c->len[i];
      if (l > 0 && l <= 16)
        { const uint32_t base = ((uint32_t) bits[i] << (16-l)) & 0xffff, span = 1u << (16-l);
          uint32_t j;
          for (j = 0; j < span && base+j < 0x10000; j++)
            c->look[base+j] = (uint8_t) i;
        }
    }
  c->have = 1;
  return 0;
}

int64_t fga_one_codec_decode(const fga_one_codec *c, const uint8_t *in, int64_t nbits, uint8_t *out, int64_t cap)
{ const int64_t nbytes = (nbits+7) >> 3;
  uint8_t *canon;
  int64_t pos, o = 0, w;
  if (nbytes > 0 && in[0] == 0xff)
    { const int64_t m = (nbits >> 3) - 1;
      if (m < 0 || m > cap) return -1;
      memcpy(out,in+1,(size_t) m);
      return m;
    }
  canon = malloc((size_t) nbytes + 8);
  if (canon == NULL) return -1;
  memcpy(canon,in,(size_t) nbytes);
  memset(canon+nbytes,0,8);
  if (nbits >= 64)
    { uint8_t x = canon[0]; canon[0] = canon[7]; canon[7] = x; }
  for (w = 0; (w+1)*64 <= nbits; w++)
    { uint8_t *b = canon + 8*w, t;
      int k;
      for (k = 0; k < 4; k++)
        { t = b[k]; b[k] = b[7-k]; b[7-k] = t; }
    }
  if (canon[0] & 0x40)
    { free(canon);
      return -1;
    }
  pos = 2;
  while (pos < nbits)
    { uint32_t look = 0;
      int k, sym, l;
      for (k = 0; k < 3; k++)
        look = (look << 8) | canon[(pos >> 3) + k];
      look = (look >> (8 - (pos & 7))) & 0xffff;
      sym = c->look[look];
      if (sym == c->esc)
        { uint32_t lit = 0;
          pos += c->esclen;
          if (c->esclen <= 0 || pos + 8 > nbits)          /* the literal byte must lie inside the stream */
            { free(canon);
              return -1;
            }
          for (k = 0; k < 2; k++)
            lit = (lit << 8) | canon[(pos >> 3) + k];
          sym = (int) ((lit >> (8 - (pos & 7))) & 0xff);
          l = 8;
        }
      else
        l = c->len[sym];
      if (l <= 0 || o >= cap)
        { free(canon);
          return -1;
        }
      pos += l;
      out[o++] = (uint8_t) sym;
    }
  free(canon);
  return o;
}

/* footer: ASCII count lines, binary '&' (index) and ';' (list code) lines, "^" */
int fga_one_footer_codecs(const uint8_t *buf, size_t size, fga_one_codec *codec)
{ int64_t off;
  const uint8_t *p, *end;
  if (size < 16) return 0;
  memcpy(&off,buf+size-8,8);
  if (off <= 0 || (size_t) off >= size-8) return 0;
  p = buf+off; end = buf+size-8;
  while (p < end && *p != '^')
    { if (!(*p & 0x80))
        { const uint8_t *e = memchr(p,'\n',(size_t) (end-p));
          if (e == NULL) return 0;
          p = e+1;
          continue;
        }
      { const uint8_t x = *p++;
        const int k = (x & 0x7f) >> 1;
        int64_t v, f0;
        int u, t;
        if (p >= end) return 1;
        t = *p++;
        if ((u = fga_one_int(p,end,&v)) == 0 || v < 0) return 1;
        p += u;
        if (k == 53)
          { if (v > 0)
              { if ((u = fga_one_int(p,end,&f0)) == 0) return 1;
                p += u;
                if (v > 1)
                  { int w;
                    if (p >= end) return 1;
                    w = *p++;
                    if (x & 1)
                      { int64_t nb;
                        if ((u = fga_one_int(p,end,&nb)) == 0 || nb < 0) return 1;
                        p += u + ((nb+7) >> 3);
                      }
                    else
                      p += (v-1)*w;
                  }
              }
          }
        else if (k == 52)
          { if ((x & 1) || p+v > end) return 1;
            if (t >= 0 && t < 128 && fga_one_codec_parse(codec+t,p,v)) return 1;
            p += v;
          }
        else
          return 1;
        if (p > end) return 1;
      }
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  line iterator
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct
  { const uint8_t *p, *end;
    uint8_t  nfld[128], fld[128][8];
    fga_one_codec codec[128];
    uint8_t *dec;  int64_t deccap;         /* decoded bytes of the current line's list     */
    int64_t *ints; int64_t intcap;         /* the current line's INT_LIST                  */
    char    *ref[4];                       /* '<' lines by their number 1..3               */
    char     ftype[16];                    /* the file type of line 1 ("aln", "gdb", ...)  */
  } one_file;

typedef struct
  { int      type;
    int64_t  ival[8]; int nint;
    const char *str; int64_t slen;
    const int64_t *list; int64_t llen;
  } one_line;

static void one_close(one_file *F)
{ int i;
  for (i = 0; i < 128; i++) free(F->codec[i].look);
  for (i = 0; i < 4; i++) free(F->ref[i]);
  free(F->dec); free(F->ints);
}

/* header up to "$": schema, references */
static int one_open(one_file *F, const uint8_t *buf, size_t size, const char *path)
{ const uint8_t *p = buf, *end = buf+size;
  int seen = 0, first = 1;
  memset(F,0,sizeof(*F));
  memset(F->nfld,0xff,sizeof(F->nfld));
  while (p < end && !seen)
    { const uint8_t *e = memchr(p,'\n',(size_t) (end-p));
      size_t n = e ? (size_t) (e-p) : (size_t) (end-p);
      char *ln = strndup((const char *) p,n);
      if (ln == NULL) { fga_set_error("out of memory"); return 1; }
      if (first)
        { int major;
          first = 0;
          if (sscanf(ln,"1 %d %15s",&major,F->ftype) != 2)
            { fga_set_error("%s is not a ONEcode file",path);
              free(ln);
              return 1;
            }
        }
      else if (ln[0] == '~' && (ln[2] == 'O' || ln[2] == 'D') && n > 6)
        { int t = (unsigned char) ln[4], k = 0, nf = 0, used = 0;
          const char *q = ln+5;
          if (sscanf(q," %d%n",&nf,&used) == 1 && nf <= 8 && t < 128)
            { q += used;
              for (k = 0; k < nf; k++)
                { int l = 0; char name[32];
                  if (sscanf(q," %d %31s%n",&l,name,&used) < 2) break;
                  q += used;
                  F->fld[t][k] = !strcmp(name,"INT") ? F_INT : !strcmp(name,"REAL") ? F_REAL :
                                 !strcmp(name,"CHAR") ? F_CHAR : !strcmp(name,"STRING") ? F_STRING :
                                 !strcmp(name,"INT_LIST") ? F_INT_LIST : !strcmp(name,"REAL_LIST") ? F_REAL_LIST :
                                 !strcmp(name,"DNA") ? F_DNA : F_STRING_LIST;
                }
              if (k == nf) F->nfld[t] = (uint8_t) nf;
            }
        }
      else if (ln[0] == '<')                      /* "< <len> <string> <number>" */
        { int len = 0, used = 0, num = 0;
          if (sscanf(ln+1," %d %n",&len,&used) >= 1 && (size_t) (1+used+len) <= n)
            { const char *s = ln+1+used;
              if (sscanf(s+len," %d",&num) == 1 && num >= 1 && num <= 3 && F->ref[num] == NULL)
                F->ref[num] = strndup(s,(size_t) len);
            }
        }
      else if (ln[0] == '$')
        { if (atoi(ln+1) != 0)
            { fga_set_error("%s is a big-endian ONEcode file",path);
              free(ln);
              return 1;
            }
          seen = 1;
        }
      free(ln);
      p += n + (e ? 1 : 0);
    }
  if (!seen)
    { fga_set_error("%s: not a binary ONEcode file (no $ line)",path);
      return 1;
    }
  if (fga_one_footer_codecs(buf,size,F->codec))
    { fga_set_error("%s: malformed ONEcode footer",path);
      return 1;
    }
  F->p = p; F->end = end;
  return 0;
}

/* 1: a line was read, 0: end of data, -1: error (message set) */
static int one_next(one_file *F, one_line *L, const char *path)
{ const uint8_t *p = F->p, *end = F->end;
  uint8_t x;
  int k, t, i;
  for (;;)
    { if (p >= end || *p == '\n')
        { F->p = p;
          return 0;
        }
      x = *p++;
      k = (x & 0x7f) >> 1;
      if (!(x & 0x80) || k > 55)
        { fga_set_error("%s: unexpected byte 0x%02x in the binary data section",path,x);
          return -1;
        }
      if (k == 55)                                /* blank line */
        continue;
      break;
    }
  t = k < 26 ? 'A'+k : k < 52 ? 'a'+(k-26) : k == 54 ? '/' : '?';
  if (t == '?' || (t != '/' && F->nfld[t] == 0xff))
    { fga_set_error("%s: line type code %d is not in the file's schema",path,k);
      return -1;
    }
  memset(L,0,sizeof(*L));
  L->type = t;
  { const int nf = (t == '/') ? 1 : F->nfld[t];
    for (i = 0; i < nf; i++)
      { const int kind = (t == '/') ? F_STRING : F->fld[t][i];
        int64_t v;
        int u;
        switch (kind)
        { case F_INT:
            if ((u = fga_one_int(p,end,&v)) == 0) goto trunc;
            p += u;
            if (L->nint < 8) L->ival[L->nint++] = v;
            break;
          case F_REAL:
            if (p+8 > end) goto trunc;
            p += 8;
            break;
          case F_CHAR:
            if (p+1 > end) goto trunc;
            p += 1;
            break;
          default:
            if ((u = fga_one_int(p,end,&v)) == 0 || v < 0) goto trunc;
            p += u;
            if (kind == F_INT_LIST)
              { int64_t f0, j;
                L->llen = v;
                if (v+1 > F->intcap)
                  { F->intcap = 2*v + 256;
                    free(F->ints);
                    F->ints = malloc(sizeof(int64_t)*(size_t) F->intcap);
                    if (F->ints == NULL) { fga_set_error("out of memory"); return -1; }
                  }
                L->list = F->ints;
                if (v > 0)
                  { if ((u = fga_one_int(p,end,&f0)) == 0) goto trunc;
                    p += u;
                    F->ints[0] = f0;
                    if (v > 1)
                      { int w;
                        const uint8_t *src;
                        if (p >= end) goto trunc;
                        w = *p++;
                        if (w < 1 || w > 8) goto trunc;
                        if (x & 1)
                          { int64_t nb, got;
                            if ((u = fga_one_int(p,end,&nb)) == 0 || nb < 0 || p+u+((nb+7)>>3) > end) goto trunc;
                            p += u;
                            if (!F->codec[t].have)
                              { fga_set_error("%s: compressed %c lines but no code for them in the footer",path,t);
                                return -1;
                              }
                            if ((v-1)*w + 8 > F->deccap)
                              { F->deccap = 2*(v-1)*w + 256;
                                free(F->dec);
                                F->dec = malloc((size_t) F->deccap);
                                if (F->dec == NULL) { fga_set_error("out of memory"); return -1; }
                              }
                            got = fga_one_codec_decode(F->codec+t,p,nb,F->dec,F->deccap);
                            if (got != (v-1)*w)
                              { fga_set_error("%s: a compressed %c line does not decode to its size",path,t);
                                return -1;
                              }
                            p += (nb+7) >> 3;
                            src = F->dec;
                          }
                        else
                          { if (p + (v-1)*w > end) goto trunc;
                            src = p;
                            p += (v-1)*w;
                          }
                        for (j = 1; j < v; j++)           /* sign-extended little-endian differences */
                          { uint64_t d = 0;
                            memcpy(&d,src + (j-1)*w,(size_t) w);
                            if (w < 8 && (d >> (8*w-1)) & 1)
                              d |= ~(uint64_t) 0 << (8*w);
                            F->ints[j] = F->ints[j-1] + (int64_t) d;
                          }
                      }
                  }
              }
            else if ((x & 1) && v > 0)
              { int64_t nb, got, need = v;
                if ((u = fga_one_int(p,end,&nb)) == 0 || nb < 0 || p+u+((nb+7)>>3) > end) goto trunc;
                p += u;
                if (kind != F_STRING)
                  { p += (nb+7) >> 3;
                    break;
                  }
                if (!F->codec[t].have)
                  { fga_set_error("%s: compressed %c lines but no code for them in the footer",path,t);
                    return -1;
                  }
                if (need+8 > F->deccap)
                  { F->deccap = 2*need + 256;
                    free(F->dec);
                    F->dec = malloc((size_t) F->deccap);
                    if (F->dec == NULL) { fga_set_error("out of memory"); return -1; }
                  }
                got = fga_one_codec_decode(F->codec+t,p,nb,F->dec,F->deccap);
                if (got != need)
                  { fga_set_error("%s: a compressed %c line does not decode to its size",path,t);
                    return -1;
                  }
                p += (nb+7) >> 3;
                L->str = (const char *) F->dec; L->slen = v;
              }
            else if (kind == F_STRING)
              { if (p+v > end) goto trunc;
                L->str = (const char *) p; L->slen = v; p += v;
              }
            else if (kind == F_REAL_LIST)
              { if (p + 8*v > end) goto trunc;
                p += 8*v;
              }
            else if (kind == F_DNA)
              { if (p + ((v+3)>>2) > end) goto trunc;
                p += (v+3)>>2;
              }
            else
              { fga_set_error("%s: string-list lines are not supported",path);
                return -1;
              }
            break;
        }
      }
  }
  F->p = p;
  return 1;

trunc:
  fga_set_error("%s: truncated or malformed binary ONEcode line",path);
  return -1;
}

/* ------------------------------------------------------------------------------------------------------------------
 *  .1aln
 * ---------------------------------------------------------------------------------------------------------------- */

static char *join_path(const char *cwd, const char *p)
{ char *r;
  if (p == NULL) return NULL;
  if (p[0] == '/' || cwd == NULL || cwd[0] == '\0')
    return strdup(p);
  if (asprintf(&r,"%s/%s",cwd,p) < 0) return NULL;
  return r;
}

/* a "$ <n>" line in the ASCII header marks the binary container */
static int has_binary_marker(const uint8_t *buf, size_t size)
{ const uint8_t *p = buf, *end = buf+size;
  while (p < end && !(*p & 0x80))
    { const uint8_t *e = memchr(p,'\n',(size_t) (end-p));
      if (p[0] == '$' && p+1 < end && p[1] == ' ')
        return 1;
      if (e == NULL) break;
      p = e+1;
    }
  return 0;
}

/* bounded number scanning for the text form: never looks beyond `e` (the buffer need not be NUL-terminated and a
   line costs its own length -- glibc's sscanf would measure the whole rest of the buffer on every call) */
static int scan_ll(const char **q, const char *e, long long *val)
{ const char *p = *q;
  int neg = 0, nd = 0;
  unsigned long long v = 0;
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
  if (p < e && (*p == '-' || *p == '+')) neg = (*p++ == '-');
  while (p < e && *p >= '0' && *p <= '9')
    { if (v > (unsigned long long) 0x7fffffffffffffffLL/10 - 1) return 0;
      v = 10*v + (unsigned) (*p++ - '0');
      nd += 1;
    }
  if (nd == 0) return 0;
  *val = neg ? -(long long) v : (long long) v;
  *q = p;
  return 1;
}

/* the text form of a .1aln: one line per ONEcode line, "<type> <fields...>", lists as "<n> v1 ... vn" */
static int read_ascii_1aln(const char *txt, size_t size, const char *path, fga_alns **out, int *tspace,
                           char **db1, char **db2)
{ const char *p = txt, *end = txt+size;
  fga_alns *R = calloc(1,sizeof(fga_alns));
  int64_t acap = 0, tcap = 0, toff = 0;
  fga_aln *cur = NULL;
  char *ref[4] = { NULL, NULL, NULL, NULL };
  int first = 1, rc = 1, i, have_t = 0, have_x = 0;

  if (R == NULL) goto oom;
  R->alns = malloc(sizeof(fga_aln)); R->tbytes = malloc(16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;
  while (p < end)
    { const char *e = memchr(p,'\n',(size_t) (end-p));
      const char *le = e ? e : end;                      /* end of this line */
      const size_t n = (size_t) (le-p);
      const char *q = p+1;
      long long v[6];
      if (first)
        { first = 0;
          q = p;
          if (!(scan_ll(&q,le,v) && v[0] == 1 && scan_ll(&q,le,v+1) && le-q >= 4 && memcmp(q," aln",4) == 0 &&
                (q+4 == le || q[4] == ' ' || q[4] == '\r')))
            { fga_set_error("%s is neither a binary nor a text .1aln",path);
              goto done;
            }
        }
      else if (n >= 1)
        switch (p[0])
        { case 't':
            if (scan_ll(&q,le,v) && tspace) *tspace = (int) v[0];
            break;
          case '<':                                       /* "< <len> <string> <number>" */
            if (scan_ll(&q,le,v) && v[0] >= 0 && q < le && (long long) (le-q-1) >= v[0])
              { const char *s = q+1, *r = q+1+v[0];
                if (scan_ll(&r,le,v+1) && v[1] >= 1 && v[1] <= 3 && ref[v[1]] == NULL)
                  ref[v[1]] = strndup(s,(size_t) v[0]);
              }
            break;
          case 'A':
            for (i = 0; i < 6; i++)
              if (!scan_ll(&q,le,v+i))
                { fga_set_error("%s: malformed A line",path);
                  goto done;
                }
            if (R->naln >= acap)
              { fga_aln *a;
                acap = 2*acap + 1024;
                a = realloc(R->alns,sizeof(fga_aln)*(size_t) acap);
                if (a == NULL) goto oom;
                R->alns = a;
              }
            cur = R->alns + R->naln++;
            memset(cur,0,sizeof(*cur));
            cur->aread = (int32_t) v[0]; cur->abpos = (int32_t) v[1]; cur->aepos = (int32_t) v[2];
            cur->bread = (int32_t) v[3]; cur->bbpos = (int32_t) v[4]; cur->bepos = (int32_t) v[5];
            cur->unit = -1; cur->seq = (int32_t) (R->naln-1); cur->toff = toff;
            have_t = have_x = 0;
            break;
          case 'R':
            if (cur) cur->flags |= 0x1;
            break;
          case 'D':
            if (cur && scan_ll(&q,le,v)) cur->diffs = (int32_t) v[0];
            break;
          case 'T': case 'X':
            if (cur != NULL)
              { const int isT = (p[0] == 'T');
                long long cnt, k;
                if (!scan_ll(&q,le,&cnt) || cnt < 0 || cnt > 0x3fffffff)
                  { fga_set_error("%s: malformed %c line of alignment %lld",path,p[0],(long long) R->naln);
                    goto done;
                  }
                if (isT ? have_t : have_x)
                  { fga_set_error("%s: alignment %lld has two %c lines",path,(long long) R->naln,p[0]);
                    goto done;
                  }
                if ((have_t || have_x) && 2*cnt != cur->tlen)
                  { fga_set_error("%s: T and X lists of alignment %lld differ in length",path,(long long) R->naln);
                    goto done;
                  }
                if (!have_t && !have_x)
                  { if (toff + 2*cnt + 16 > tcap)
                      { uint8_t *b;
                        tcap = 2*(toff + 2*cnt) + 4096;
                        b = realloc(R->tbytes,(size_t) tcap);
                        if (b == NULL) goto oom;
                        R->tbytes = b;
                      }
                    memset(R->tbytes+toff,0,(size_t) (2*cnt));
                    cur->tlen = (int32_t) (2*cnt);
                    toff += 2*cnt;
                    R->ntrace = toff;
                  }
                if (isT) have_t = 1; else have_x = 1;
                for (k = 0; k < cnt; k++)
                  { long long val;
                    if (!scan_ll(&q,le,&val))
                      { fga_set_error("%s: a %c line of alignment %lld holds fewer than its %lld values",path,p[0],
                                      (long long) R->naln,cnt);
                        goto done;
                      }
                    if (val < 0 || val > 255)
                      { fga_set_error("%s: trace value %lld of alignment %lld does not fit the 8-bit trace form",path,
                                      val,(long long) R->naln);
                        goto done;
                      }
                    R->tbytes[cur->toff + 2*k + (isT ? 1 : 0)] = (uint8_t) val;
                  }
              }
            break;
          default:
            break;
        }
      p = le + (e ? 1 : 0);
    }
  if (first)
    { fga_set_error("%s is empty",path);
      goto done;
    }
  if (tspace && *tspace > 125)
    { fga_set_error("%s: trace spacing %d > 125 means 16-bit traces, which this reader does not hold",path,*tspace);
      goto done;
    }
  if (db1) *db1 = join_path(ref[3],ref[1]);
  if (db2) *db2 = join_path(ref[3],ref[2]);
  rc = 0;
  goto done;

oom:
  fga_set_error("out of memory reading %s",path);
done:
  for (i = 0; i < 4; i++) free(ref[i]);
  if (rc != 0)
    { fga_alns_free(R);
      return 1;
    }
  *out = R;
  return 0;
}

int fga_read_1aln(const char *path, fga_alns **out, int *tspace, char **db1, char **db2)
{ FILE *f = fopen(path,"rb");
  uint8_t *buf = NULL;
  long size;
  one_file F;
  one_line L;
  fga_alns *R = NULL;
  int64_t acap = 0, tcap = 0, toff = 0;
  int64_t *xlist = NULL; int64_t xcap = 0, xlen = -1;
  int rc = 1, st, have_t = 0, opened = 0;
  fga_aln *cur = NULL;

  *out = NULL;
  if (tspace) *tspace = 100;
  if (db1) *db1 = NULL;
  if (db2) *db2 = NULL;
  if (f == NULL)
    { fga_set_error("cannot open %s",path);
      return 1;
    }
  if (fseek(f,0,SEEK_END) != 0 || (size = ftell(f)) < 32 || fseek(f,0,SEEK_SET) != 0 ||
      (buf = malloc((size_t) size)) == NULL || fread(buf,1,(size_t) size,f) != (size_t) size)
    { fga_set_error("cannot read %s",path);
      goto done;
    }
  if (!has_binary_marker(buf,(size_t) size))      /* the text form (ours with FGA_ALN_ASCII=1, or ONEview's output) */
    { rc = read_ascii_1aln((const char *) buf,(size_t) size,path,out,tspace,db1,db2);
      free(buf);
      fclose(f);
      return rc;
    }
  opened = 1;
  if (one_open(&F,buf,(size_t) size,path)) goto done;
  if (strcmp(F.ftype,"aln") != 0)
    { fga_set_error("%s is a ONEcode .%s file, not a .1aln",path,F.ftype);
      goto done;
    }
  R = calloc(1,sizeof(fga_alns));
  if (R == NULL) goto oom;
  R->alns = malloc(sizeof(fga_aln)); R->tbytes = malloc(16);
  if (R->alns == NULL || R->tbytes == NULL) goto oom;

  while ((st = one_next(&F,&L,path)) == 1)
    switch (L.type)
    { case 't':
        if (tspace && L.nint > 0) *tspace = (int) L.ival[0];
        break;
      case 'A':
        if (L.nint < 6)
          { fga_set_error("%s: malformed A line",path);
            goto done;
          }
        if (R->naln >= acap)
          { fga_aln *a;
            acap = 2*acap + 1024;
            a = realloc(R->alns,sizeof(fga_aln)*(size_t) acap);
            if (a == NULL) goto oom;
            R->alns = a;
          }
        cur = R->alns + R->naln++;
        memset(cur,0,sizeof(*cur));
        cur->aread = (int32_t) L.ival[0]; cur->abpos = (int32_t) L.ival[1]; cur->aepos = (int32_t) L.ival[2];
        cur->bread = (int32_t) L.ival[3]; cur->bbpos = (int32_t) L.ival[4]; cur->bepos = (int32_t) L.ival[5];
        cur->unit = -1; cur->seq = (int32_t) (R->naln-1); cur->toff = toff;
        have_t = 0; xlen = -1;
        break;
      case 'R':
        if (cur) cur->flags |= 0x1;
        break;
      case 'D':
        if (cur && L.nint > 0) cur->diffs = (int32_t) L.ival[0];
        break;
      case 'T': case 'X':
        if (cur == NULL) break;
        if (L.type == 'X' && !have_t)             /* X before T: keep it until T arrives */
          { if (L.llen > xcap)
              { xcap = 2*L.llen + 64;
                free(xlist);
                xlist = malloc(sizeof(int64_t)*(size_t) xcap);
                if (xlist == NULL) goto oom;
              }
            memcpy(xlist,L.list,sizeof(int64_t)*(size_t) L.llen);
            xlen = L.llen;
            break;
          }
        { int64_t n = L.llen, i;
          for (i = 0; i < n; i++)
            if (L.list[i] < 0 || L.list[i] > 255)
              { fga_set_error("%s: trace value %lld of alignment %lld does not fit the 8-bit trace form",path,
                              (long long) L.list[i],(long long) R->naln);
                goto done;
              }
          if (n > 0x3fffffff || (L.type == 'T' && have_t))
            { fga_set_error("%s: malformed %c line of alignment %lld",path,L.type,(long long) R->naln);
              goto done;
            }
          if (L.type == 'T')
            { if (toff + 2*n + 16 > tcap)
                { uint8_t *b;
                  tcap = 2*(toff + 2*n) + 4096;
                  b = realloc(R->tbytes,(size_t) tcap);
                  if (b == NULL) goto oom;
                  R->tbytes = b;
                }
              for (i = 0; i < n; i++)
                { R->tbytes[toff+2*i] = 0; R->tbytes[toff+2*i+1] = (uint8_t) L.list[i]; }
              cur->tlen = (int32_t) (2*n);
              have_t = 1;
              if (xlen >= 0 && xlen != n)
                { fga_set_error("%s: T and X lists of alignment %lld differ in length",path,(long long) R->naln);
                  goto done;
                }
              if (xlen == n)
                for (i = 0; i < n; i++)
                  { if (xlist[i] < 0 || xlist[i] > 255)
                      { fga_set_error("%s: trace value %lld of alignment %lld does not fit the 8-bit trace form",path,
                                      (long long) xlist[i],(long long) R->naln);
                        goto done;
                      }
                    R->tbytes[toff+2*i] = (uint8_t) xlist[i];
                  }
              toff += 2*n;
              R->ntrace = toff;
            }
          else                                    /* X after T */
            { if (2*n != cur->tlen)
                { fga_set_error("%s: T and X lists of alignment %lld differ in length",path,(long long) R->naln);
                  goto done;
                }
              for (i = 0; i < n; i++)
                R->tbytes[cur->toff+2*i] = (uint8_t) L.list[i];
            }
        }
        break;
      default:            /* skeleton (g S G C), chains (a p), L Q E Z U, comments */
        break;
    }
  if (st < 0) goto done;
  if (db1) *db1 = join_path(F.ref[3],F.ref[1]);
  if (db2) *db2 = join_path(F.ref[3],F.ref[2]);
  rc = 0;
  goto done;

oom:
  fga_set_error("out of memory reading %s",path);
done:
  if (opened) one_close(&F);
  free(buf); free(xlist);
  fclose(f);
  if (rc != 0)
    { fga_alns_free(R);
      return 1;
    }
  *out = R;
  return 0;
}
