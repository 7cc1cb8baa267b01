/* fga_aln.c -- .1aln writers (binary and ASCII ONEcode).
 *
 * Replaces open_Aln_Write / Write_Aln_Overlap / Write_Aln_Trace (reference alncode.c:239-305) and Write_Skeleton
 * (GDB.c:2065-2092) as used by la_merge (FastGA.c:4049-4113).  The reference writes the *binary* ONEcode form
 * through its vendored ONElib; fga_write_1aln_binary emits that container natively (same schema alncode.c:19-52,
 * same line content and order), so the reference's seeking readers (ALNtoPAF, ALNshow, ...: oneGoto needs the object
 * index of the binary footer) take the file as they take their own.  fga_write_1aln writes the equivalent ASCII
 * ONEcode file, which `ONEview` prints identically apart from the '!' provenance and '<' path lines.
 *   body:  t <tspace> / g + skeleton of genome 1 / g + skeleton of genome 2 (not for self) /
 *          per alignment: A aread abpos aepos bread bbpos bepos / R (complement) / D diffs /
 *                         T n b-deltas (odd trace bytes) / X n diffs (even trace bytes)
 * An ASCII file must embed its schema ('~' lines) and its count lines ('#', '@', '+', '%') ahead of the data, so the
 * statistics are accumulated first and the body is written second; a binary file carries them in its footer.
 *
 * Binary ONEcode as written here (the container format of ONElib.c, restated):
 *   ASCII header: "1 3 aln 2 1", '!' provenance, '<' references, the '~' schema lines, "$ 0" (little endian)
 *   data lines:   1 type byte 0x80 | k << 1  (k = letter index: 'A'..'Z' 0..25, 'a'..'z' 26..51; bit 0 = list codec
 *                 in use, never set here), then the fields: INT and list lengths as variable-length "ltf" integers
 *                 (0..63 in one byte 0x40|v, up to 8191 in two bytes 0x20|hi,lo, else a byte count n-1 and the n low
 *                 bytes), STRING bytes raw, INT_LIST as ltf(first), one byte w, then the successive differences in w
 *                 bytes each (w = the smallest signed width that holds every difference)
 *   '\n' end of data; footer: per line type in schema order its ASCII count lines ('#' count, '@' max list, '+' list
 *                 total, '%' per-object maxima) and, for object types (g, S, A), a binary '&' line holding the byte
 *                 offset of the data start and of every object; "^\n"; the footer's own offset as 8 raw bytes.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/stat.h>

#include "fga_host.h"
#include "fastga_amd.h"

/* a new output file in place of an old one of the same name: a plain file nobody else links to is removed first -- giving an
   existing file's blocks back inside fopen(.., "w") costs 2-5 ms on the boxes' ext4 (journal + discard), a run that writes its
   result over the last one's pays that inside the comparison's span; the unlink of a file still in the page cache 0.03 ms.
   Anything else (a pipe, a device, a symbolic or hard link) is opened as it is */
static FILE *fopen_fresh(const char *path)
{ struct stat sb;
  if (lstat(path,&sb) == 0 && S_ISREG(sb.st_mode) && sb.st_nlink == 1)
    unlink(path);
  return fopen(path,"w");
}

static const char *ALN_SCHEMA_LINES =
  "~ D t 1 3 INT                 trace point spacing in a - global\n"
  "~ O g 0                       groups scaffolds into a GDB skeleton\n"
  "~ G S 0                         collection of scaffolds constituting a GDB\n"
  "~ O S 1 6 STRING              id for a scaffold\n"
  "~ D G 1 3 INT                 gap of given length\n"
  "~ D C 1 3 INT                 contig of given length\n"
  "~ O a 0                       groups A's into a colinear chain\n"
  "~ G A 0                         chains (a) group alignment objects (A)\n"
  "~ D p 2 3 INT 3 INT           spacing in a,b between end of previous alignment and start of next\n"
  "~ O A 6 3 INT 3 INT 3 INT 3 INT 3 INT 3 INT\n"
  "~ D L 2 3 INT 3 INT           lengths of sequences a and b\n"
  "~ D R 0                       flag: reverse-complement sequence b\n"
  "~ D D 1 3 INT                 differences: number of diffs = substitions + indels\n"
  "~ D T 1 8 INT_LIST            trace points in b\n"
  "~ D X 1 8 INT_LIST            number of differences in alignment per trace interval\n"
  "~ D Q 1 3 INT                 quality: alignment confidence in phred units (currently unused)\n"
  "~ D E 1 3 INT                 match: number of equal bases (currently unused)\n"
  "~ D Z 1 6 STRING              cigar string: encodes precise alignment (currently unused)\n"
  "~ D U 1 3 INT                 putative unit size of a TR alignment (FASTAN)\n";

typedef struct
  { int64_t nS, maxS, totS, nG, nC;           /* over all skeletons written     */
    int64_t gC, gG, gS, gSt;                  /* per-g maxima                    */
    int64_t sC, sG;                           /* per-S maxima                    */
  } skel_stats;

/* threads the record formatters use: 8 unless the caller's thread has asked for another number (the pipeline passes its
 * -T); the bytes written do not depend on it (contiguous record ranges, concatenated in order) */
#define WRITER_MAXT 64
static __thread int writer_threads = 8;
void fga_aln_writer_threads(int n)
{ writer_threads = n < 1 ? 1 : (n > WRITER_MAXT ? WRITER_MAXT : n); }

static void skeleton_stats(const fga_gdb *G, skel_stats *st)
{ int s, c;
  int64_t nC = 0, nG = 0, tS = 0;
  for (s = 0; s < G->nscaff; s++)
    { int64_t spos = 0, sc = 0, sg = 0;
      int64_t hl = strlen(G->headers + G->scaffolds[s].hoff);
      if (hl > st->maxS) st->maxS = hl;
      tS += hl;
      for (c = G->scaffolds[s].fctg; c < G->scaffolds[s].ectg; c++)
        { if (G->contigs[c].sbeg > spos)
            sg += 1;
          sc += 1;
          spos = G->contigs[c].sbeg + G->contigs[c].clen;
        }
      if (G->scaffolds[s].slen > spos)
        sg += 1;
      nC += sc; nG += sg;
      if (sc > st->sC) st->sC = sc;
      if (sg > st->sG) st->sG = sg;
    }
  st->nS += G->nscaff; st->totS += tS; st->nC += nC; st->nG += nG;
  if (nC > st->gC) st->gC = nC;
  if (nG > st->gG) st->gG = nG;
  if (G->nscaff > st->gS) st->gS = G->nscaff;
  if (tS > st->gSt) st->gSt = tS;
}

static void write_skeleton(FILE *f, const fga_gdb *G)
{ int s, c;
  fprintf(f,"g\n");
  for (s = 0; s < G->nscaff; s++)
    { const char *head = G->headers + G->scaffolds[s].hoff;
      int64_t spos = 0;
      fprintf(f,"S %d %s\n",(int) strlen(head),head);
      for (c = G->scaffolds[s].fctg; c < G->scaffolds[s].ectg; c++)
        { if (G->contigs[c].sbeg > spos)
            fprintf(f,"G %lld\n",(long long) (G->contigs[c].sbeg - spos));
          fprintf(f,"C %lld\n",(long long) G->contigs[c].clen);
          spos = G->contigs[c].sbeg + G->contigs[c].clen;
        }
      if (G->scaffolds[s].slen > spos)
        fprintf(f,"G %lld\n",(long long) (G->scaffolds[s].slen - spos));
    }
}

typedef struct
  { const fga_alns *A;
    int64_t i0, i1;
    char   *buf;
    size_t  len;
    int     fail;
  } fmt_job;

#define PUT_INT(v)                                                     \
    { unsigned _u; int _n = 0; char _t[12]; int _v = (v);              \
      if (_v < 0) { buf[len++] = '-'; _u = (unsigned) (-(long long) _v); } else _u = (unsigned) _v; \
      do { _t[_n++] = (char) ('0' + _u % 10); _u /= 10; } while (_u);  \
      while (_n) buf[len++] = _t[--_n];                                \
    }

/* A / R / D / T / X lines of records [i0,i1) */
static void *fmt_thread(void *arg)
{ fmt_job *J = arg;
  const fga_alns *A = J->A;
  int64_t i;
  size_t cap = 64, len = 0;
  char *buf;
  for (i = J->i0; i < J->i1; i++)
    cap += 160 + 4*(size_t) A->alns[i].tlen;        /* <= 4 chars per trace byte incl. the separator */
  buf = malloc(cap);
  if (buf == NULL)
    { J->fail = 1;
      return NULL;
    }
  for (i = J->i0; i < J->i1; i++)
    { const fga_aln *a = A->alns+i;
      const uint8_t *tr = A->tbytes + a->toff;
      int x, q;
      int fld[6];
      fld[0] = a->aread; fld[1] = a->abpos; fld[2] = a->aepos; fld[3] = a->bread; fld[4] = a->bbpos; fld[5] = a->bepos;
      buf[len++] = 'A';
      for (q = 0; q < 6; q++)
        { buf[len++] = ' ';
          PUT_INT(fld[q])
        }
      buf[len++] = '\n';
      if (a->flags & 1)
        { buf[len++] = 'R'; buf[len++] = '\n'; }
      buf[len++] = 'D'; buf[len++] = ' ';
      PUT_INT(a->diffs)
      buf[len++] = '\n';
      for (q = 1; q >= 0; q--)
        { buf[len++] = q ? 'T' : 'X'; buf[len++] = ' ';
          PUT_INT(a->tlen/2)
          for (x = q; x < a->tlen; x += 2)
            { unsigned v = tr[x];
              buf[len++] = ' ';
              if (v >= 100) { buf[len++] = (char) ('0' + v/100); v %= 100; buf[len++] = (char) ('0' + v/10); buf[len++] = (char) ('0' + v%10); }
              else if (v >= 10) { buf[len++] = (char) ('0' + v/10); buf[len++] = (char) ('0' + v%10); }
              else buf[len++] = (char) ('0' + v);
            }
          buf[len++] = '\n';
        }
    }
  J->buf = buf; J->len = len;
  return NULL;
}

int fga_write_1aln(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *A, int tspace,
                   const char *db1_name, const char *db2_name, const char *command_line)
{ FILE *f;
  skel_stats st;
  int64_t i, nR = 0, maxT = 0, totT = 0;
  char date[64], *cwd;
  time_t t = time(NULL);
  char *obuf;

  memset(&st,0,sizeof(st));
  skeleton_stats(g1,&st);
  if (g2 != NULL)
    skeleton_stats(g2,&st);
  for (i = 0; i < A->naln; i++)
    { int64_t tl = A->alns[i].tlen/2;
      if (A->alns[i].flags & 1) nR += 1;
      if (tl > maxT) maxT = tl;
      totT += tl;
    }

  f = fopen_fresh(path);
  if (f == NULL)
    { fga_set_error("cannot open %s for writing",path);
      return 1;
    }
  obuf = malloc(1<<22);
  if (obuf != NULL)
    setvbuf(f,obuf,_IOFBF,1<<22);
  strftime(date,sizeof(date),"%Y-%m-%d_%H:%M:%S",localtime(&t));
  cwd = getcwd(NULL,0);

  fprintf(f,"1 3 aln 2 1\n");
  fprintf(f,"! 4 6 FastGA 3 0.1 %d %s %d %s\n",(int) strlen(command_line),command_line,(int) strlen(date),date);
  fprintf(f,".\n");
  fprintf(f,"< %d %s 1\n",(int) strlen(db1_name),db1_name);
  if (g2 != NULL && db2_name != NULL)
    fprintf(f,"< %d %s 2\n",(int) strlen(db2_name),db2_name);
  if (cwd != NULL)
    fprintf(f,"< %d %s 3\n",(int) strlen(cwd),cwd);
  free(cwd);
  fprintf(f,".\n");
  fputs(ALN_SCHEMA_LINES,f);
  fprintf(f,".\n");

  fprintf(f,"# t 1\n");
  fprintf(f,"# g %d\n",g2 != NULL ? 2 : 1);
  fprintf(f,"%% g # C %lld\n",(long long) st.gC);
  if (st.gG > 0) fprintf(f,"%% g # G %lld\n",(long long) st.gG);
  fprintf(f,"%% g # S %lld\n",(long long) st.gS);
  fprintf(f,"%% g + S %lld\n",(long long) st.gSt);
  fprintf(f,"# S %lld\n",(long long) st.nS);
  fprintf(f,"@ S %lld\n",(long long) st.maxS);
  fprintf(f,"+ S %lld\n",(long long) st.totS);
  fprintf(f,"%% S # C %lld\n",(long long) st.sC);
  if (st.sG > 0) fprintf(f,"%% S # G %lld\n",(long long) st.sG);
  if (st.nG > 0) fprintf(f,"# G %lld\n",(long long) st.nG);
  fprintf(f,"# C %lld\n",(long long) st.nC);
  if (A->naln > 0)
    { fprintf(f,"# A %lld\n",(long long) A->naln);
      fprintf(f,"%% A # D 1\n");
      if (nR > 0) fprintf(f,"%% A # R 1\n");
      fprintf(f,"%% A # T 1\n");
      fprintf(f,"%% A + T %lld\n",(long long) maxT);
      fprintf(f,"%% A # X 1\n");
      fprintf(f,"%% A + X %lld\n",(long long) maxT);
      if (nR > 0) fprintf(f,"# R %lld\n",(long long) nR);
      fprintf(f,"# D %lld\n",(long long) A->naln);
      fprintf(f,"# T %lld\n",(long long) A->naln);
      fprintf(f,"@ T %lld\n",(long long) maxT);
      fprintf(f,"+ T %lld\n",(long long) totT);
      fprintf(f,"# X %lld\n",(long long) A->naln);
      fprintf(f,"@ X %lld\n",(long long) maxT);
      fprintf(f,"+ X %lld\n",(long long) totT);
    }
  fprintf(f,".\n");

  fprintf(f,"t %d\n",tspace);
  write_skeleton(f,g1);
  if (g2 != NULL)
    write_skeleton(f,g2);

  { /* the record body dominates the file: records are formatted by hand, in parallel (contiguous record ranges
       into per-thread buffers sized from the trace lengths), and written in order */
    int nth = writer_threads, t;
    fmt_job  job[WRITER_MAXT];
    pthread_t th[WRITER_MAXT];
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    if (nc > 0 && nc < nth) nth = (int) nc;
    if (totT < 200000) nth = 1;
    for (t = 0; t < nth; t++)
      { job[t].A = A;
        job[t].i0 = (A->naln*t)/nth; job[t].i1 = (A->naln*(t+1))/nth;
        job[t].buf = NULL; job[t].len = 0; job[t].fail = 0;
      }
    for (t = 1; t < nth; t++)
      if (pthread_create(th+t,NULL,fmt_thread,job+t) != 0)
        { fmt_thread(job+t); th[t] = 0; }
    fmt_thread(job);
    for (t = 1; t < nth; t++)
      if (th[t] != 0)
        pthread_join(th[t],NULL);
    for (t = 0; t < nth; t++)
      { if (job[t].fail)
          { int q;
            for (q = 0; q < nth; q++) free(job[q].buf);
            fga_set_error("out of memory");
            fclose(f);
            free(obuf);
            return 1;
          }
      }
    fflush(f);
    for (t = 0; t < nth; t++)
      { if (job[t].len > 0)
          fwrite(job[t].buf,1,job[t].len,f);
        free(job[t].buf);
      }
  }
  if (fclose(f) != 0)
    { fga_set_error("IO error writing %s",path);
      free(obuf);
      return 1;
    }
  free(obuf);
  return 0;
}


/* ------------------------------------------------------------------------------------------------------------
 * binary ONEcode
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct { uint8_t *p; size_t len, cap; int fail; } bbuf;

static void bb_need(bbuf *B, size_t n)
{ if (B->len + n > B->cap)
    { size_t nc = B->cap*2 + n + 4096;
      uint8_t *q = realloc(B->p,nc);
      if (q == NULL) { B->fail = 1; return; }
      B->p = q; B->cap = nc;
    }
}

static inline int ltf_put(uint8_t *u, int64_t val)        /* ONElib's integer packing (intPut), <= 9 bytes */
{ if (val >= 0)
    { if (val < 0x40)   { u[0] = (uint8_t) (val | 0x40); return 1; }
      if (val < 0x2000) { u[0] = (uint8_t) ((val >> 8) | 0x20); u[1] = (uint8_t) (val & 0xff); return 2; }
      { int n = 2; uint64_t v = (uint64_t) val;
        while (n < 8 && (v >> (8*n)) != 0) n += 1;
        u[0] = (uint8_t) (n-1);
        memcpy(u+1,&v,8);
        return n+1;
      }
    }
  if (val >= -0x40) { u[0] = (uint8_t) (val | 0x40); return 1; }
  { int n = 2; uint64_t v = (uint64_t) val;
    while (n < 8 && (~v >> (8*n)) != 0) n += 1;
    u[0] = (uint8_t) (0x80 | (n-1));
    memcpy(u+1,&v,8);
    return n+1;
  }
}

#define TYPE_BYTE(t) ((uint8_t) (0x80 | ((((t) >= 'a') ? 26 + ((t)-'a') : ((t)-'A')) << 1)))
#define INDEX_BYTE   ((uint8_t) (0x80 | (53 << 1)))                  /* '&' */

static void bb_type(bbuf *B, char t)
{ bb_need(B,1); if (B->fail) return;
  B->p[B->len++] = TYPE_BYTE(t);
}
static void bb_int(bbuf *B, int64_t v)
{ bb_need(B,16); if (B->fail) return;
  B->len += ltf_put(B->p+B->len,v);
}
static void bb_bytes(bbuf *B, const void *src, size_t n)
{ bb_need(B,n); if (B->fail) return;
  memcpy(B->p+B->len,src,n); B->len += n;
}

/* ---- list codes: ONElib compresses the difference bytes of long INT_LIST streams with a Huffman code per line type
 *      (trained by the writer, stored in a ';' footer line, ONElib.c:2412-2506, 3305-3336, 3470-3575).  Here the code of
 *      a line type is built from ALL its difference bytes (the whole set is in memory), limited to 16 bits, canonical. */
typedef struct
  { int      have;
    uint8_t  len[256];
    uint16_t bits[256];
  } wcodec;

static void wcodec_build(const uint64_t *hist, wcodec *C)
{ uint64_t f[256];
  int i, nsym = 0, maxlen;
  memset(C,0,sizeof(*C));
  for (i = 0; i < 256; i++)
    { f[i] = hist[i];
      nsym += (hist[i] > 0);
    }
  if (nsym == 0)
    return;
  do
    { uint64_t w[512];
      int parent[512], alive[512], n = 0, na, a, b;
      for (i = 0; i < 256; i++)
        if (f[i] > 0)
          { w[n] = f[i]; parent[n] = -1; alive[n] = 1; n += 1; }
      na = n;
      while (na > 1)                               /* plain Huffman: join the two lightest live nodes */
        { a = b = -1;
          for (i = 0; i < n; i++)
            if (alive[i])
              { if (a < 0 || w[i] < w[a]) { b = a; a = i; }
                else if (b < 0 || w[i] < w[b]) b = i;
              }
          w[n] = w[a]+w[b]; parent[n] = -1; alive[n] = 1;
          parent[a] = parent[b] = n; alive[a] = alive[b] = 0;
          n += 1; na -= 1;
        }
      maxlen = 0;
      { int k = 0;
        for (i = 0; i < 256; i++)
          if (f[i] > 0)
            { int l = 0, q = k++;
              while (parent[q] >= 0) { q = parent[q]; l += 1; }
              if (l == 0) l = 1;                   /* a single symbol still needs one bit */
              C->len[i] = (uint8_t) (l > 255 ? 255 : l);
              if (l > maxlen) maxlen = l;
            }
          else
            C->len[i] = 0;
      }
      if (maxlen > 16)                             /* flatten the distribution and try again */
        for (i = 0; i < 256; i++)
          if (f[i] > 0) f[i] = (f[i] >> 1) + 1;
    }
  while (maxlen > 16);
  { uint32_t code = 0;
    int l;
    for (l = 1; l <= 16; l++)                      /* canonical codes: by length, then by symbol */
      { for (i = 0; i < 256; i++)
          if (C->len[i] == l)
            C->bits[i] = (uint16_t) code++;
        code <<= 1;
      }
  }
  C->have = 1;
}

/* the code in ONElib's serialised form (vcSerialize): endian byte, escape code (-1: none) and length, then per symbol
 * its length and, if non-zero, its right-aligned code */
static size_t wcodec_blob(const wcodec *C, uint8_t *out)
{ uint8_t *o = out;
  int esc = -1, esclen = 0, i;
  *o++ = 0;
  memcpy(o,&esc,4); o += 4;
  memcpy(o,&esclen,4); o += 4;
  for (i = 0; i < 256; i++)
    { *o++ = C->len[i];
      if (C->len[i] > 0)
        { memcpy(o,C->bits+i,2); o += 2; }
    }
  return (size_t) (o-out);
}

/* n bytes -> code stream in ONElib's layout (vcEncode): two zero flag bits, codes most significant bit first, 64-bit
 * words stored little endian, the last partial word byte by byte, bytes 0 and 7 exchanged once the stream has a full
 * word.  Returns the bit count, or -1 when the stream would not be smaller than the n bytes. */
static int64_t wcodec_encode(const wcodec *C, const uint8_t *in, int n, uint8_t *out)
{ int64_t nbits = 2, w;
  int i;
  for (i = 0; i < n; i++)
    { if (C->len[in[i]] == 0) return -1;
      nbits += C->len[in[i]];
    }
  if (nbits >= 8*(int64_t) n)
    return -1;
  memset(out,0,(size_t) ((nbits+7) >> 3) + 8);
  { int64_t pos = 2;
    for (i = 0; i < n; i++)
      { const int l = C->len[in[i]];
        uint32_t c = (uint32_t) C->bits[in[i]] << (24-l);        /* code left-aligned in 24 bits */
        const int sh = (int) (pos & 7);
        uint8_t *o = out + (pos >> 3);
        c >>= sh;                                                /* l <= 16, sh <= 7: fits 24 bits */
        o[0] |= (uint8_t) (c >> 16); o[1] |= (uint8_t) (c >> 8); o[2] |= (uint8_t) c;
        pos += l;
      }
  }
  for (w = 0; (w+1)*64 <= nbits; w++)
    { uint8_t *b = out + 8*w, t;
      int k;
      for (k = 0; k < 4; k++)
        { t = b[k]; b[k] = b[7-k]; b[7-k] = t; }
    }
  if (nbits >= 64)
    { uint8_t t = out[0]; out[0] = out[7]; out[7] = t; }
  return nbits;
}

/* INT_LIST line of type t with the n values v[0], v[stride], ...: type byte (bit 0: compressed), length, first value,
 * difference width, then the differences -- as they are, or as a code stream preceded by its bit count */
static void bb_listline(bbuf *B, char t, const uint8_t *v, int n, int stride, const wcodec *C, uint8_t *tmp)
{ int i, w = 1;
  size_t tpos;
  bb_need(B,1); if (B->fail) return;
  tpos = B->len;
  B->p[B->len++] = TYPE_BYTE(t);
  bb_int(B,n);
  if (n <= 0) return;
  bb_int(B,v[0]);
  if (n == 1) return;
  for (i = 1; i < n; i++)
    { int d = (int) v[i*stride] - (int) v[(i-1)*stride];
      if (d >= 128 || d < -128) { w = 2; break; }
    }
  if (w == 1 && C != NULL && C->have && n > 8)
    { uint8_t *d = tmp, *e = tmp + n;
      int64_t nbits;
      for (i = 1; i < n; i++)
        d[i-1] = (uint8_t) ((int) v[i*stride] - (int) v[(i-1)*stride]);
      nbits = wcodec_encode(C,d,n-1,e);
      if (nbits > 0)
        { bb_need(B,12 + (size_t) ((nbits+7) >> 3)); if (B->fail) return;
          B->p[tpos] |= 1;
          B->p[B->len++] = 1;
          B->len += ltf_put(B->p+B->len,nbits);
          memcpy(B->p+B->len,e,(size_t) ((nbits+7) >> 3));
          B->len += (size_t) ((nbits+7) >> 3);
          return;
        }
    }
  bb_need(B,1 + (size_t) (n-1)*w); if (B->fail) return;
  B->p[B->len++] = (uint8_t) w;
  for (i = 1; i < n; i++)
    { int d = (int) v[i*stride] - (int) v[(i-1)*stride];
      B->p[B->len++] = (uint8_t) (d & 0xff);
      if (w == 2) B->p[B->len++] = (uint8_t) ((d >> 8) & 0xff);
    }
}

/* the '&' index line of object type t: offsets[0..n] (n+1 >= 2 values) */
static void bb_index(bbuf *B, char t, const int64_t *off, int64_t n)
{ int64_t i; uint64_t mask = 0; int w;
  bb_need(B,2); if (B->fail) return;
  B->p[B->len++] = INDEX_BYTE;
  B->p[B->len++] = (uint8_t) t;
  bb_int(B,n+1);
  bb_int(B,off[0]);
  for (i = 1; i <= n; i++)
    { int64_t d = off[i]-off[i-1];
      mask |= (uint64_t) (d >= 0 ? d : -(d+1));
    }
  mask >>= 7;
  for (w = 1; w < 8 && mask != 0; w++)
    mask >>= 8;
  bb_need(B,1 + (size_t) n*w); if (B->fail) return;
  B->p[B->len++] = (uint8_t) w;
  for (i = 1; i <= n; i++)
    { int64_t d = off[i]-off[i-1];
      memcpy(B->p+B->len,&d,w);            /* little endian: the w low bytes */
      B->len += w;
    }
}

static void bin_skeleton(bbuf *B, const fga_gdb *G, int64_t base, int64_t *goff, int64_t *ng, int64_t *soff, int64_t *ns)
{ int s, c;
  goff[++(*ng)] = base + (int64_t) B->len;
  bb_type(B,'g');
  for (s = 0; s < G->nscaff; s++)
    { const char *head = G->headers + G->scaffolds[s].hoff;
      int64_t spos = 0;
      size_t hl = strlen(head);
      soff[++(*ns)] = base + (int64_t) B->len;
      bb_type(B,'S'); bb_int(B,(int64_t) hl); bb_bytes(B,head,hl);
      for (c = G->scaffolds[s].fctg; c < G->scaffolds[s].ectg; c++)
        { if (G->contigs[c].sbeg > spos)
            { bb_type(B,'G'); bb_int(B,G->contigs[c].sbeg - spos); }
          bb_type(B,'C'); bb_int(B,G->contigs[c].clen);
          spos = G->contigs[c].sbeg + G->contigs[c].clen;
        }
      if (G->scaffolds[s].slen > spos)
        { bb_type(B,'G'); bb_int(B,G->scaffolds[s].slen - spos); }
    }
}

typedef struct
  { const fga_alns *A;
    int64_t i0, i1;
    bbuf    B;
    int64_t *rel;            /* start of record i relative to the job's buffer */
    const wcodec *ct, *cx;   /* list codes of the T and X lines (NULL: none)     */
    uint8_t *tmp;            /* 3 * (longest list) + 32 bytes                    */
  } bin_job;

static void *bin_thread(void *arg)
{ bin_job *J = arg;
  const fga_alns *A = J->A;
  int64_t i;
  for (i = J->i0; i < J->i1 && !J->B.fail; i++)
    { const fga_aln *a = A->alns+i;
      const uint8_t *tr = A->tbytes + a->toff;
      J->rel[i-J->i0] = (int64_t) J->B.len;
      bb_type(&J->B,'A');
      bb_int(&J->B,a->aread); bb_int(&J->B,a->abpos); bb_int(&J->B,a->aepos);
      bb_int(&J->B,a->bread); bb_int(&J->B,a->bbpos); bb_int(&J->B,a->bepos);
      if (a->flags & 1)
        bb_type(&J->B,'R');
      bb_type(&J->B,'D'); bb_int(&J->B,a->diffs);
      bb_listline(&J->B,'T',tr+1,a->tlen/2,2,J->ct,J->tmp);
      bb_listline(&J->B,'X',tr,a->tlen/2,2,J->cx,J->tmp);
    }
  return NULL;
}

/* The file as a stream: header + skeletons when it is opened, records appended set by set in final order (each set formatted
 * on the writer threads and written through the one stream), footer -- counts in schema order, the object indices -- when it is
 * closed.  A comparison that runs phase 2 in several parts writes a part's records while the next part's kernels run
 * (fga_pipeline.c); fga_write_1aln_binary is open + one append + close. */
struct fga_aln_stream
  { FILE    *f;
    char    *path;
    const fga_gdb *g1, *g2;
    skel_stats st;
    int64_t  base, pos;          /* start of the data / absolute position of the next record */
    int64_t *goff, *soff, *aoff; /* object indices; aoff grows */
    int64_t  ng, ns, na, acap;
    int64_t  nR, maxT, totT;
    int64_t  datalen;            /* bytes of data written behind the header */
    wcodec   ct, cx;
    int      trained;
    int      fail;
  };

/* a record set formatted for a stream, not yet in it: the formatter threads' buffers, every record's start relative to its
   buffer, and what the footer counts.  Made by any thread (fga_aln_stream_format reads the stream, it does not change it),
   committed in the file's order */
struct fga_aln_block
  { int      nth;
    bin_job  job[WRITER_MAXT];
    int64_t  naln, nR, maxT, totT;
  };

void fga_aln_block_free(fga_aln_block *K)
{ int q;
  if (K == NULL) return;
  for (q = 0; q < WRITER_MAXT; q++) { free(K->job[q].B.p); free(K->job[q].rel); free(K->job[q].tmp); }
  free(K);
}

/* list codes for the T and X lines, like the reference's files carry them once a type has > ~100 KB of list data.
 * Opt-in (FGA_ALN_CODEC=1): training + encoding doubles the writer's time (it is inside the timed hot path) for a
 * file 2.4x smaller; from 16 K trace points on, trained on the first set that is appended */
static int stream_wants_codecs(void)
{ return getenv("FGA_ALN_CODEC") != NULL && atoi(getenv("FGA_ALN_CODEC")) != 0; }

static void stream_train(fga_aln_stream *S, const fga_alns *A)
{ int64_t i, totT = 0;
  if (S->trained) return;
  S->trained = 1;
  for (i = 0; i < A->naln; i++) totT += A->alns[i].tlen/2;
  if (totT >= 16384 && stream_wants_codecs())
    { uint64_t ht[256], hx[256];
      int64_t k;
      memset(ht,0,sizeof(ht)); memset(hx,0,sizeof(hx));
      for (i = 0; i < A->naln; i++)
        { const uint8_t *tr = A->tbytes + A->alns[i].toff;
          const int64_t n = A->alns[i].tlen/2;
          int okt = 1, okx = 1;
          for (k = 1; k < n && (okt || okx); k++)
            { const int dt = (int) tr[2*k+1] - (int) tr[2*k-1], dx = (int) tr[2*k] - (int) tr[2*k-2];
              if (dt >= 128 || dt < -128) okt = 0;
              if (dx >= 128 || dx < -128) okx = 0;
            }
          for (k = 1; k < n; k++)
            { if (okt) ht[(uint8_t) ((int) tr[2*k+1] - (int) tr[2*k-1])] += 1;
              if (okx) hx[(uint8_t) ((int) tr[2*k] - (int) tr[2*k-2])] += 1;
            }
        }
      wcodec_build(ht,&S->ct);
      wcodec_build(hx,&S->cx);
    }
}

/* can a set be formatted before its turn in the file has come (fga_aln_stream_format on the thread that made it, the commit
   when the sets before it are in)?  Not when list codes are wanted: they are trained on the first set of the FILE */
int fga_aln_stream_preformats(const fga_aln_stream *S)
{ return S != NULL && !S->fail && !stream_wants_codecs(); }

int fga_aln_stream_format(const fga_aln_stream *S, const fga_alns *A, fga_aln_block **out)
{ fga_aln_block *K;
  pthread_t th[WRITER_MAXT];
  int nth = writer_threads, q;
  int64_t i, maxT = 0, totT = 0, nR = 0;
  if (out != NULL) *out = NULL;
  if (S == NULL || A == NULL || out == NULL)
    { fga_set_error("fga_aln_stream_format: null argument");
      return 1;
    }
  K = calloc(1,sizeof(fga_aln_block));
  if (K == NULL) goto oom;
  for (i = 0; i < A->naln; i++)
    { int64_t tl = A->alns[i].tlen/2;
      if (A->alns[i].flags & 1) nR += 1;
      if (tl > maxT) maxT = tl;
      totT += tl;
    }
  K->naln = A->naln; K->nR = nR; K->maxT = maxT; K->totT = totT;
  { long nc = sysconf(_SC_NPROCESSORS_ONLN);
    if (nc > 0 && nc < nth) nth = (int) nc;
    if (totT < 200000) nth = 1;
  }
  K->nth = nth;
  for (q = 0; q < nth; q++)
    { bin_job *J = K->job + q;
      J->A = A;
      J->i0 = (A->naln*q)/nth; J->i1 = (A->naln*(q+1))/nth;
      J->rel = malloc(sizeof(int64_t)*(J->i1-J->i0+1));
      J->tmp = malloc((size_t) (3*maxT + 64));
      J->ct = S->ct.have ? &S->ct : NULL; J->cx = S->cx.have ? &S->cx : NULL;
      if (J->rel == NULL || J->tmp == NULL) goto oom;
    }
  { int started[WRITER_MAXT];
    for (q = 1; q < nth; q++)
      { started[q] = pthread_create(th+q,NULL,bin_thread,K->job+q) == 0;
        if (!started[q]) bin_thread(K->job+q);
      }
    bin_thread(K->job);
    for (q = 1; q < nth; q++)
      if (started[q])
        pthread_join(th[q],NULL);
  }
  for (q = 0; q < nth; q++)
    { if (K->job[q].B.fail) goto oom;
      K->job[q].A = NULL;                       /* (the set may go once its records are formatted) */
      free(K->job[q].tmp); K->job[q].tmp = NULL;
    }
  *out = K;
  return 0;
oom:
  fga_set_error("out of memory");
  fga_aln_block_free(K);
  return 1;
}

/* the block's records behind those already in the stream; the block is consumed */
int fga_aln_stream_commit(fga_aln_stream *S, fga_aln_block *K)
{ int q, rc = 1;
  int64_t i;
  if (S == NULL || K == NULL || S->fail)
    { fga_set_error("fga_aln_stream_commit: no open stream");
      fga_aln_block_free(K);
      return 1;
    }
  if (S->na + K->naln + 2 > S->acap)
    { const int64_t nc = 2*(S->na + K->naln) + 1024;
      int64_t *x = realloc(S->aoff,sizeof(int64_t)*nc);
      if (x == NULL)
        { fga_set_error("out of memory");
          goto done;
        }
      S->aoff = x; S->acap = nc;
    }
  S->nR += K->nR; S->totT += K->totT;
  if (K->maxT > S->maxT) S->maxT = K->maxT;
  for (q = 0; q < K->nth; q++)
    { const bin_job *J = K->job + q;
      for (i = J->i0; i < J->i1; i++)
        S->aoff[S->na + i + 1] = S->pos + J->rel[i-J->i0];
      S->pos += (int64_t) J->B.len;
    }
  S->na += K->naln;
  /* One stream writes the file.  Measured in round 5 on the 289 MB of a 3 Gbp comparison's 4.2 M records (32 formatter
     threads): this 108-166 ms; every job writing its own stretch with pwrite 137-211 ms (buffered writes to one file take
     turns on the inode's lock); the jobs copying into a shared mapping of the file 420-530 ms (a page fault per 4 KB). */
  { int ok = 1;
    for (q = 0; q < K->nth; q++)
      if (K->job[q].B.len > 0)
        { ok &= fwrite(K->job[q].B.p,1,K->job[q].B.len,S->f) == K->job[q].B.len;
          S->datalen += (int64_t) K->job[q].B.len;
        }
    if (!ok)
      { fga_set_error("IO error writing %s",S->path);
        goto done;
      }
  }
  rc = 0;
done:
  fga_aln_block_free(K);
  if (rc) S->fail = 1;
  return rc;
}

int fga_aln_stream_open(const char *path, const fga_gdb *g1, const fga_gdb *g2, int tspace,
                        const char *db1_name, const char *db2_name, const char *command_line, fga_aln_stream **out)
{ fga_aln_stream *S = calloc(1,sizeof(fga_aln_stream));
  char date[64], *cwd;
  time_t t = time(NULL);
  bbuf H, B;
  int64_t nscaf;
  *out = NULL;
  memset(&H,0,sizeof(H)); memset(&B,0,sizeof(B));
  if (S == NULL) goto oom;
  S->g1 = g1; S->g2 = g2;
  S->path = strdup(path);
  skeleton_stats(g1,&S->st);
  if (g2 != NULL)
    skeleton_stats(g2,&S->st);
  nscaf = g1->nscaff + (g2 != NULL ? g2->nscaff : 0);
  S->goff = malloc(sizeof(int64_t)*4);
  S->soff = malloc(sizeof(int64_t)*(nscaf+2));
  S->acap = 1024;
  S->aoff = malloc(sizeof(int64_t)*S->acap);
  if (S->path == NULL || S->goff == NULL || S->soff == NULL || S->aoff == NULL) goto oom;

  /* ---- ASCII header ---- */
  strftime(date,sizeof(date),"%Y-%m-%d_%H:%M:%S",localtime(&t));
  cwd = getcwd(NULL,0);
  { char *hdr = NULL;
    int   n;
    n = asprintf(&hdr,"1 3 aln 2 1\n! 4 6 FastGA 3 0.1 %d %s %d %s\n.\n< %d %s 1\n",
                 (int) strlen(command_line),command_line,(int) strlen(date),date,(int) strlen(db1_name),db1_name);
    if (n < 0) { free(cwd); goto oom; }
    bb_bytes(&H,hdr,(size_t) n); free(hdr);
    if (g2 != NULL && db2_name != NULL)
      { n = asprintf(&hdr,"< %d %s 2\n",(int) strlen(db2_name),db2_name);
        if (n < 0) { free(cwd); goto oom; }
        bb_bytes(&H,hdr,(size_t) n); free(hdr);
      }
    if (cwd != NULL)
      { n = asprintf(&hdr,"< %d %s 3\n",(int) strlen(cwd),cwd);
        if (n < 0) { free(cwd); goto oom; }
        bb_bytes(&H,hdr,(size_t) n); free(hdr);
      }
    free(cwd);
    bb_bytes(&H,".\n",2);
    bb_bytes(&H,ALN_SCHEMA_LINES,strlen(ALN_SCHEMA_LINES));
    bb_bytes(&H,"$ 0\n",4);
  }
  if (H.fail) goto oom;
  S->base = (int64_t) H.len;                    /* the data starts here: index[0] of every object type */
  S->goff[0] = S->soff[0] = S->aoff[0] = S->base;

  /* ---- data: t, skeletons ---- */
  bb_type(&B,'t'); bb_int(&B,tspace);
  bin_skeleton(&B,g1,S->base,S->goff,&S->ng,S->soff,&S->ns);
  if (g2 != NULL)
    bin_skeleton(&B,g2,S->base,S->goff,&S->ng,S->soff,&S->ns);
  if (B.fail) goto oom;
  S->pos = S->base + (int64_t) B.len;
  S->datalen = (int64_t) B.len;

  S->f = fopen_fresh(path);
  if (S->f == NULL)
    { fga_set_error("cannot open %s for writing",path);
      goto fail;
    }
  if (fwrite(H.p,1,H.len,S->f) != H.len || fwrite(B.p,1,B.len,S->f) != B.len)
    { fga_set_error("IO error writing %s",path);
      goto fail;
    }
  free(H.p); free(B.p);
  *out = S;
  return 0;
oom:
  fga_set_error("out of memory");
fail:
  free(H.p); free(B.p);
  if (S != NULL)
    { if (S->f != NULL) fclose(S->f);
      free(S->path); free(S->goff); free(S->soff); free(S->aoff); free(S);
    }
  return 1;
}

int fga_aln_stream_append(fga_aln_stream *S, const fga_alns *A)
{ if (S == NULL || A == NULL || S->fail)
    { fga_set_error("fga_aln_stream_append: no open stream");
      return 1;
    }
  if (A->naln == 0)
    return 0;
  { fga_aln_block *K;
    stream_train(S,A);
    if (fga_aln_stream_format(S,A,&K))
      { S->fail = 1;
        return 1;
      }
    return fga_aln_stream_commit(S,K);
  }
}

int64_t fga_aln_stream_records(const fga_aln_stream *S) { return S == NULL ? 0 : S->na; }

int fga_aln_stream_close(fga_aln_stream *S, int keep)
{ bbuf F;
  int rc = 1;
  if (S == NULL) return 0;
  memset(&F,0,sizeof(F));
  if (!keep || S->fail)
    goto done;
  /* ---- footer: counts in schema order (t g S G C a A p L R D T X ...), '&' index lines of g, S, A ---- */
#define FPRINT(...) { char *_s = NULL; int _n = asprintf(&_s,__VA_ARGS__); if (_n < 0) goto oom; bb_bytes(&F,_s,(size_t) _n); free(_s); }
  FPRINT("# t 1\n")
  FPRINT("# g %lld\n",(long long) S->ng)
  FPRINT("%% g # C %lld\n",(long long) S->st.gC)
  if (S->st.gG > 0) FPRINT("%% g # G %lld\n",(long long) S->st.gG)
  FPRINT("%% g # S %lld\n",(long long) S->st.gS)
  FPRINT("%% g + S %lld\n",(long long) S->st.gSt)
  bb_index(&F,'g',S->goff,S->ng);
  FPRINT("# S %lld\n",(long long) S->st.nS)
  FPRINT("@ S %lld\n",(long long) S->st.maxS)
  FPRINT("+ S %lld\n",(long long) S->st.totS)
  FPRINT("%% S # C %lld\n",(long long) S->st.sC)
  if (S->st.sG > 0) FPRINT("%% S # G %lld\n",(long long) S->st.sG)
  bb_index(&F,'S',S->soff,S->ns);
  if (S->st.nG > 0) FPRINT("# G %lld\n",(long long) S->st.nG)
  FPRINT("# C %lld\n",(long long) S->st.nC)
  if (S->na > 0)
    { FPRINT("# A %lld\n",(long long) S->na)
      FPRINT("%% A # D 1\n")
      if (S->nR > 0) FPRINT("%% A # R 1\n")
      FPRINT("%% A # T 1\n")
      if (S->maxT > 0) FPRINT("%% A + T %lld\n",(long long) S->maxT)
      FPRINT("%% A # X 1\n")
      if (S->maxT > 0) FPRINT("%% A + X %lld\n",(long long) S->maxT)
      bb_index(&F,'A',S->aoff,S->na);
      if (S->nR > 0) FPRINT("# R %lld\n",(long long) S->nR)
      FPRINT("# D %lld\n",(long long) S->na)
      FPRINT("# T %lld\n",(long long) S->na)
      if (S->maxT > 0) FPRINT("@ T %lld\n",(long long) S->maxT)
      if (S->totT > 0) FPRINT("+ T %lld\n",(long long) S->totT)
      FPRINT("# X %lld\n",(long long) S->na)
      if (S->maxT > 0) FPRINT("@ X %lld\n",(long long) S->maxT)
      if (S->totT > 0) FPRINT("+ X %lld\n",(long long) S->totT)
      { const wcodec *cc[2] = { &S->ct, &S->cx };           /* ';' lines: CHAR line type, STRING serialised code */
        int z;
        for (z = 0; z < 2; z++)
          if (cc[z]->have)
            { uint8_t blob[1024];
              const size_t bl = wcodec_blob(cc[z],blob);
              bb_need(&F,2); if (F.fail) goto oom;
              F.p[F.len++] = (uint8_t) (0x80 | (52 << 1));
              F.p[F.len++] = (uint8_t) (z == 0 ? 'T' : 'X');
              bb_int(&F,(int64_t) bl);
              bb_bytes(&F,blob,bl);
            }
      }
    }
  FPRINT("^\n")
#undef FPRINT
  if (F.fail) goto oom;
  { const int64_t foot = S->base + S->datalen + 1;       /* + the end-of-data '\n' */
    int ok = 1;
    ok &= fputc('\n',S->f) != EOF;
    ok &= fwrite(F.p,1,F.len,S->f) == F.len;
    ok &= fwrite(&foot,sizeof(int64_t),1,S->f) == 1;
    if (!ok)
      { fga_set_error("IO error writing %s",S->path);
        goto done;
      }
  }
  rc = 0;
  goto done;
oom:
  fga_set_error("out of memory");
done:
  if (S->f != NULL && fclose(S->f) != 0 && rc == 0)
    { fga_set_error("IO error writing %s",S->path);
      rc = 1;
    }
  if (rc != 0 || !keep)
    { unlink(S->path);
      if (!keep && !S->fail) rc = 0;
    }
  free(F.p);
  free(S->path); free(S->goff); free(S->soff); free(S->aoff); free(S);
  return rc;
}

int fga_write_1aln_binary(const char *path, const fga_gdb *g1, const fga_gdb *g2, const fga_alns *A, int tspace,
                          const char *db1_name, const char *db2_name, const char *command_line)
{ fga_aln_stream *S;
  if (fga_aln_stream_open(path,g1,g2,tspace,db1_name,db2_name,command_line,&S))
    return 1;
  if (fga_aln_stream_append(S,A))
    { char e[512];
      snprintf(e,sizeof(e),"%s",fga_last_error());
      fga_aln_stream_close(S,0);
      fga_set_error("%s",e);
      return 1;
    }
  return fga_aln_stream_close(S,1);
}
