/* fga_gdb.c -- genome database (GDB) reader and FASTA -> GDB producer.
 *
 * Format contract (reference GDB.c):
 *   skeleton  <root>.gdb   ASCII ONEcode, schema of GDB.c:37-47: 'f' 4 base frequencies, per scaffold
 *                          'S <name>' followed by 'G <gap>' / 'C <contig length>' lines in order
 *                          (reader: GDB.c:1313-1355; writer: GDB.c:1589-1614).
 *   bases     .<root>.bps  every contig starts on a byte boundary, base i sits in bits 2*(i&3) of byte
 *                          i>>2 (GDB.c:960-976, gene_core.c:372-398), a=0 c=1 g=2 t=3.
 * The reference's Read_GDB tries <root>.1gdb first and then <root>.gdb (GDB.c:1198-1224); this module
 * reads the ASCII form (what it also writes).  The binary .1gdb form needs the ONEcode binary codec and
 * is reported as unsupported with a clear message.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <ctype.h>
#include <time.h>
#include <unistd.h>
#include <sys/time.h>
#include <pthread.h>
#include <zlib.h>

#include "fga_host.h"

/* ------------------------------------------------------------------------------------------------ */

static __thread char Error_Msg[1024];

void fga_set_error(const char *fmt, ...)
{ va_list ap;
  va_start(ap,fmt);
  vsnprintf(Error_Msg,sizeof(Error_Msg),fmt,ap);
  va_end(ap);
}

const char *fga_last_error(void)
{ return Error_Msg; }

double fga_wall(void)
{ struct timeval tv;
  gettimeofday(&tv,NULL);
  return tv.tv_sec + 1e-6*tv.tv_usec;
}

char *fga_path_dir(const char *path)
{ const char *s = strrchr(path,'/');
  if (s == NULL)
    return strdup(".");
  if (s == path)
    return strdup("/");
  return strndup(path,s-path);
}

char *fga_path_root(const char *path, const char *suffix)
{ const char *s = strrchr(path,'/');
  char *r;
  size_t n, m;
  s = (s == NULL) ? path : s+1;
  r = strdup(s);
  if (suffix != NULL)
    { n = strlen(r);
      m = strlen(suffix);
      if (n >= m && strcmp(r+(n-m),suffix) == 0)
        r[n-m] = '\0';
    }
  return r;
}

/* strip one of the known skeleton / source extensions from a path */
static char *strip_gdb_ext(const char *path)
{ static const char *ext[] = { ".1gdb", ".gdb", ".gix", NULL };
  char *r = strdup(path);
  size_t n = strlen(r);
  int i;
  for (i = 0; ext[i] != NULL; i++)
    { size_t m = strlen(ext[i]);
      if (n > m && strcmp(r+(n-m),ext[i]) == 0)
        { r[n-m] = '\0';
          break;
        }
    }
  return r;
}

/* ------------------------------------------------------------------------------------------------
 *  Skeleton writer (ASCII ONEcode).  Layout mirrors what the reference's ONEview prints for a .1gdb,
 *  which its own ASCII reader accepts: header, provenance, reference, embedded schema, counts, data.
 * ------------------------------------------------------------------------------------------------ */

static const char *GDB_SCHEMA_LINES =
  "~ D f 4 4 REAL 4 REAL 4 REAL 4 REAL   global: base frequency vector\n"
  "~ D u 0                               global: upper case when displayed (Deprecated)\n"
  "~ O S 1 6 STRING                      id for a scaffold\n"
  "~ D G 1 3 INT                         gap of given length\n"
  "~ D C 1 3 INT                         contig of given length\n"
  "~ D M 1 8 INT_LIST                    mask pair list for a contig\n";

static int write_skeleton_file(const fga_gdb *gdb, const char *path, const char *prog,
                               const char *command)
{ FILE *f;
  int   s, c;
  int64_t ngap, maxs, tots, maxc, maxg;
  char  date[64];
  time_t t = time(NULL);

  f = fopen(path,"w");
  if (f == NULL)
    { fga_set_error("cannot open %s for writing",path);
      return 1;
    }
  strftime(date,sizeof(date),"%Y-%m-%d_%H:%M:%S",localtime(&t));

  ngap = 0;
  maxs = 0;
  tots = 0;
  maxc = maxg = 0;
  for (s = 0; s < gdb->nscaff; s++)
    { int64_t spos = 0, sg = 0;
      int64_t hl = strlen(gdb->headers + gdb->scaffolds[s].hoff);
      if (hl > maxs) maxs = hl;
      tots += hl;
      for (c = gdb->scaffolds[s].fctg; c < gdb->scaffolds[s].ectg; c++)
        { if (gdb->contigs[c].sbeg > spos)
            sg += 1;
          spos = gdb->contigs[c].sbeg + gdb->contigs[c].clen;
        }
      if (gdb->scaffolds[s].slen > spos)
        sg += 1;
      ngap += sg;
      if (sg > maxg) maxg = sg;
      if (gdb->scaffolds[s].ectg - gdb->scaffolds[s].fctg > maxc)
        maxc = gdb->scaffolds[s].ectg - gdb->scaffolds[s].fctg;
    }

  fprintf(f,"1 3 gdb 2 1\n");
  fprintf(f,"! 4 %d %s 3 0.1 %d %s %d %s\n",(int) strlen(prog),prog,(int) strlen(command),command,
            (int) strlen(date),date);
  fprintf(f,".\n");
  fprintf(f,"< %d %s 1\n",(int) strlen(gdb->srcpath),gdb->srcpath);
  fprintf(f,".\n");
  fputs(GDB_SCHEMA_LINES,f);
  fprintf(f,".\n");
  fprintf(f,"# f 1\n");
  fprintf(f,"# S %d\n",gdb->nscaff);
  fprintf(f,"@ S %lld\n",(long long) maxs);
  fprintf(f,"+ S %lld\n",(long long) tots);
  fprintf(f,"%% S # C %lld\n",(long long) maxc);          /* per-scaffold maxima, as ONElib accumulates them */
  if (ngap > 0)
    { fprintf(f,"%% S # G %lld\n",(long long) maxg);
      fprintf(f,"# G %lld\n",(long long) ngap);
    }
  fprintf(f,"# C %d\n",gdb->ncontig);
  fprintf(f,".\n");
  fprintf(f,"f %f %f %f %f\n",gdb->freq[0],gdb->freq[1],gdb->freq[2],gdb->freq[3]);
  for (s = 0; s < gdb->nscaff; s++)
    { const char *head = gdb->headers + gdb->scaffolds[s].hoff;
      int64_t spos = 0;
      fprintf(f,"S %d %s\n",(int) strlen(head),head);
      for (c = gdb->scaffolds[s].fctg; c < gdb->scaffolds[s].ectg; c++)
        { if (gdb->contigs[c].sbeg > spos)
            fprintf(f,"G %lld\n",(long long) (gdb->contigs[c].sbeg - spos));
          fprintf(f,"C %lld\n",(long long) gdb->contigs[c].clen);
          spos = gdb->contigs[c].sbeg + gdb->contigs[c].clen;
        }
      if (gdb->scaffolds[s].slen > spos)
        fprintf(f,"G %lld\n",(long long) (gdb->scaffolds[s].slen - spos));
    }
  if (fclose(f) != 0)
    { fga_set_error("IO error writing %s",path);
      return 1;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 *  FASTA (optionally gzip'd) -> <root>.gdb + .<root>.bps        (semantics of Create_GDB, GDB.c:800-1090)
 *    - a scaffold per '>' header, name = header text after leading white space
 *    - every run of non-acgt symbols of length >= ncut (ncut = 0: every run) splits the scaffold into
 *      contigs separated by a gap; shorter runs are stored as 'a'
 *    - base frequencies = counts / total bases, stored as float
 * ------------------------------------------------------------------------------------------------ */

typedef struct
  { uint8_t *buf;
    int64_t  len, cap;
  } bytevec;

static int bv_push(bytevec *v, uint8_t x)
{ if (v->len >= v->cap)
    { v->cap = (int64_t) (1.5*v->cap) + (1<<20);
      v->buf = realloc(v->buf,v->cap);
      if (v->buf == NULL)
        return 1;
    }
  v->buf[v->len++] = x;
  return 0;
}

int fga_fasta_to_gdb(const char *fasta, const char *target, int ncut)
{ gzFile   in;
  fga_gdb  G;
  bytevec  bps = { NULL, 0, 0 };
  char    *hdr = NULL;
  int64_t  hdrcap = 0;
  int      sctop = 0, cttop = 0;
  int64_t  count[4] = { 0, 0, 0, 0 };
  int      number[256];
  static char *line = NULL;
  size_t   linecap = 1<<20;
  int      status = 1;
  char    *dir = NULL, *root = NULL, *noext = NULL, *bpath = NULL, *gpath = NULL;

  int      m;           /* bit offset inside the byte being filled          */
  uint8_t  byte;
  int64_t  clen, spos, nin;
  int      inscaf;
  /* soft mask = lower-case runs (Create_GDB keeps them as an implicit .1ano, GDB.c:988-1010) */
  int64_t  mrun = -1, nmask = 0, mcap = 0;
  int64_t *mctg = NULL, *mbeg = NULL, *mend = NULL;

  memset(&G,0,sizeof(G));

  { int i;
    for (i = 0; i < 256; i++) number[i] = 4;
    number['a'] = number['A'] = 0;
    number['c'] = number['C'] = 1;
    number['g'] = number['G'] = 2;
    number['t'] = number['T'] = 3;
  }

  in = gzopen(fasta,"r");
  if (in == NULL)
    { fga_set_error("cannot open FASTA %s",fasta);
      return 1;
    }
  gzbuffer(in,1<<20);
  line = malloc(linecap);
  if (line == NULL)
    goto oom;

  m = 0; byte = 0; clen = 0; spos = 0; nin = 0; inscaf = 0;

#define PUSH_MASK(b_,e_)                                                          \
  { if (nmask >= mcap)                                                            \
      { mcap = mcap*2 + 1024;                                                     \
        mctg = realloc(mctg,sizeof(int64_t)*mcap);                                \
        mbeg = realloc(mbeg,sizeof(int64_t)*mcap);                                \
        mend = realloc(mend,sizeof(int64_t)*mcap);                                \
        if (mctg == NULL || mbeg == NULL || mend == NULL) goto oom;               \
      }                                                                           \
    mctg[nmask] = G.ncontig; mbeg[nmask] = (b_); mend[nmask] = (e_); nmask += 1;  \
  }

#define END_CONTIG(force)                                                         \
  { if (clen > 0 || (force))                                                      \
      { if (mrun >= 0) { PUSH_MASK(mrun,clen) mrun = -1; }                        \
        if (m > 0) { if (bv_push(&bps,byte)) goto oom; }                          \
        byte = 0; m = 0;                                                          \
        if (G.ncontig >= cttop)                                                   \
          { cttop = (int) (1.2*G.ncontig) + 1000;                                 \
            G.contigs = realloc(G.contigs,sizeof(fga_contig)*(cttop+1));          \
            if (G.contigs == NULL) goto oom;                                      \
          }                                                                       \
        G.contigs[G.ncontig].clen = clen;                                         \
        G.contigs[G.ncontig].sbeg = spos;                                         \
        G.contigs[G.ncontig].boff = bps.len - ((clen+3)>>2);                      \
        G.contigs[G.ncontig].scaf = G.nscaff-1;                                   \
        G.ncontig += 1;                                                           \
        if (clen > G.maxctg) G.maxctg = clen;                                     \
        G.seqtot += clen;                                                         \
        spos += clen;                                                             \
        clen = 0;                                                                 \
      }                                                                           \
  }

#define END_SCAFFOLD()                                                            \
  { if (inscaf)                                                                   \
      { END_CONTIG(0);                                                            \
        nin = 0;                        /* N's that end a scaffold are not recorded (GDB.c:856-859, 947-950) */ \
        if (spos == 0)                                                            \
          { fga_set_error("missing sequence entry for scaffold %d of %s",G.nscaff,fasta); \
            goto fail;                                                            \
          }                                                                       \
        G.scaffolds[G.nscaff-1].ectg = G.ncontig;                                 \
        G.scaffolds[G.nscaff-1].slen = spos;                                      \
      }                                                                           \
  }

  while (gzgets(in,line,linecap) != NULL)
    { size_t len = strlen(line);
      while (len == linecap-1 && line[len-1] != '\n')      /* very long line: grow and continue */
        { linecap *= 2;
          line = realloc(line,linecap);
          if (line == NULL) goto oom;
          if (gzgets(in,line+len,linecap-len) == NULL) break;
          len += strlen(line+len);
        }
      while (len > 0 && (line[len-1] == '\n' || line[len-1] == '\r'))
        line[--len] = '\0';

      if (line[0] == '>')
        { size_t s;
          END_SCAFFOLD();
          for (s = 1; s < len; s++)
            if (!isspace((unsigned char) line[s]))
              break;
          if (G.nscaff >= sctop)
            { sctop = (int) (1.2*G.nscaff) + 500;
              G.scaffolds = realloc(G.scaffolds,sizeof(fga_scaffold)*sctop);
              if (G.scaffolds == NULL) goto oom;
            }
          if (G.hdrtot + (int64_t) (len-s) + 1 > hdrcap)
            { hdrcap = (int64_t) (1.2*(G.hdrtot+(len-s)+1)) + 10000;
              hdr = realloc(hdr,hdrcap);
              if (hdr == NULL) goto oom;
            }
          G.scaffolds[G.nscaff].fctg = G.ncontig;
          G.scaffolds[G.nscaff].hoff = G.hdrtot;
          memcpy(hdr+G.hdrtot,line+s,len-s);
          G.hdrtot += len-s;
          hdr[G.hdrtot++] = '\0';
          G.nscaff += 1;
          inscaf = 1;
          spos = 0; clen = 0; nin = 0; m = 0; byte = 0;
          continue;
        }
      if (!inscaf)
        { fga_set_error("first header of FASTA %s is missing",fasta);
          goto fail;
        }
      { size_t s;
        for (s = 0; s < len; s++)
          { int x = number[(unsigned char) line[s]];
            if (x >= 4)
              { nin += 1;
                continue;
              }
            if (nin > 0)
              { if (nin < ncut)                  /* short run: kept as a's (GDB.c:880-896) */
                  { int64_t k;
                    for (k = 0; k < nin; k++)
                      { if (m == 6) { if (bv_push(&bps,byte)) goto oom; byte = 0; m = 0; }
                        else m += 2;
                      }
                    clen += nin;
                    count[0] += nin;
                  }
                else                             /* a run at the scaffold start leaves a zero-length contig */
                  { END_CONTIG(1);
                    spos += nin;
                  }
                nin = 0;
              }
            if (line[s] >= 96)                      /* lower case: inside a soft-mask run */
              { if (mrun < 0) mrun = clen; }
            else if (mrun >= 0)
              { PUSH_MASK(mrun,clen)
                mrun = -1;
              }
            byte |= (uint8_t) (x << m);
            if (m == 6) { if (bv_push(&bps,byte)) goto oom; byte = 0; m = 0; }
            else m += 2;
            count[x] += 1;
            clen += 1;
          }
      }
    }
  END_SCAFFOLD();
  gzclose(in);
  in = NULL;

  if (G.ncontig == 0)
    { fga_set_error("FASTA %s holds no sequence",fasta);
      goto fail;
    }

  G.headers = hdr;
  G.freq[0] = (1.*count[0])/G.seqtot;
  G.freq[1] = (1.*count[1])/G.seqtot;
  G.freq[2] = (1.*count[2])/G.seqtot;
  G.freq[3] = (1.*count[3])/G.seqtot;

  { char *rp = realpath(fasta,NULL);
    G.srcpath = rp ? rp : strdup(fasta);
  }

  noext = strip_gdb_ext(target);
  dir   = fga_path_dir(noext);
  root  = fga_path_root(noext,NULL);
  if (asprintf(&bpath,"%s/.%s.bps",dir,root) < 0 || asprintf(&gpath,"%s/%s.gdb",dir,root) < 0)
    goto oom;

  { FILE *b = fopen(bpath,"w");
    if (b == NULL)
      { fga_set_error("cannot open %s for writing",bpath);
        goto fail;
      }
    if (bps.len > 0 && fwrite(bps.buf,1,bps.len,b) != (size_t) bps.len)
      { fga_set_error("IO error writing %s",bpath);
        fclose(b);
        goto fail;
      }
    fclose(b);
  }

  { char *mpath = NULL;             /* soft-mask side file (only when the FASTA mixes cases) */
    if (asprintf(&mpath,"%s/.%s.msk",dir,root) < 0) goto oom;
    if (nmask > 0 && nmask < G.seqtot)
      { FILE *mf = fopen(mpath,"w");
        int64_t hdr[2];
        if (mf == NULL)
          { fga_set_error("cannot open %s for writing",mpath);
            free(mpath);
            goto fail;
          }
        hdr[0] = G.ncontig; hdr[1] = nmask;
        fwrite(hdr,sizeof(int64_t),2,mf);
        fwrite(mctg,sizeof(int64_t),nmask,mf);
        fwrite(mbeg,sizeof(int64_t),nmask,mf);
        fwrite(mend,sizeof(int64_t),nmask,mf);
        fclose(mf);
      }
    else
      unlink(mpath);
    free(mpath);
  }

  { char cmd[2048];
    snprintf(cmd,sizeof(cmd),"fga_fasta_to_gdb %s %s",fasta,target);
    if (write_skeleton_file(&G,gpath,"FAtoGDB",cmd))
      goto fail;
  }
  status = 0;
  goto done;

oom:
  fga_set_error("out of memory creating GDB for %s",fasta);
fail:
  status = 1;
done:
  if (in != NULL) gzclose(in);
  free(line); line = NULL;
  free(mctg); free(mbeg); free(mend);
  free(bps.buf);
  free(hdr);
  free(G.scaffolds);
  free(G.contigs);
  free(G.srcpath);
  free(noext); free(dir); free(root); free(bpath); free(gpath);
  return status;
}

/* the bases of a genome (0.75 GB for 3 Gbp) from the page cache into a fresh buffer: the copy and the first touch of the
 * buffer's pages by eight threads with pread (one fread: 0.44 s; this: 0.1 s -- it is a third of what opening a 3 Gbp
 * session costs beside the two index builds) */
typedef struct { int fd; uint8_t *dst; int64_t beg, end; int bad; } image_job;

static void *image_thread(void *arg)
{ image_job *J = arg;
  int64_t at = J->beg;
  while (at < J->end)
    { const ssize_t r = pread(J->fd,J->dst+at,(size_t) (J->end-at),(off_t) at);
      if (r <= 0) { J->bad = 1; break; }
      at += r;
    }
  return NULL;
}

static int read_image(FILE *f, uint8_t *dst, int64_t len)
{ image_job job[8];
  pthread_t th[8];
  int nt = len >= ((int64_t) 64 << 20) ? 8 : 1, t, bad = 0;
  if (nt == 1)
    return fread(dst,1,(size_t) len,f) != (size_t) len;
  for (t = 0; t < nt; t++)
    { job[t].fd = fileno(f); job[t].dst = dst; job[t].beg = len*t/nt; job[t].end = len*(t+1)/nt; job[t].bad = 0; }
  { int started[8];
    for (t = 1; t < nt; t++)
      { started[t] = pthread_create(th+t,NULL,image_thread,job+t) == 0;
        if (!started[t]) image_thread(job+t);
      }
    image_thread(job);
    for (t = 1; t < nt; t++)
      if (started[t]) pthread_join(th[t],NULL);
  }
  for (t = 0; t < nt; t++)
    bad |= job[t].bad;
  return bad;
}

/* ------------------------------------------------------------------------------------------------
 *  Skeleton reader (ASCII or binary ONEcode) + whole-image load of the .bps   (Read_GDB, GDB.c:1181-1410)
 * ------------------------------------------------------------------------------------------------ */

typedef struct
  { fga_gdb *G;
    int      sctop, cttop;
    int64_t  hdrcap, boff, spos;
    /* a .1ano file only: its M lines (scaffold, beg, end) */
    int      want_m;
    int64_t *mrec, nm, mcap;
  } skel_ctx;

static int skel_mline(skel_ctx *X, int64_t scf, int64_t beg, int64_t end)
{ if (X->nm >= X->mcap)
    { int64_t *nr;
      X->mcap = X->mcap ? 2*X->mcap : 4096;
      nr = realloc(X->mrec,sizeof(int64_t)*3*X->mcap);
      if (nr == NULL) { fga_set_error("out of memory"); return 1; }
      X->mrec = nr;
    }
  X->mrec[3*X->nm] = scf; X->mrec[3*X->nm+1] = beg; X->mrec[3*X->nm+2] = end;
  X->nm += 1;
  return 0;
}

/* one skeleton line: 'f' (r4), 'S' (str,len), 'G' / 'C' (ival), '<' (str,len); everything else is ignored
 * (provenance, schema, counts, deprecated u and M lines) */
static int skel_line(skel_ctx *X, char type, const char *str, int64_t len, int64_t ival, const double *r4)
{ fga_gdb *G = X->G;
  switch (type)
  { case '<':
      if (G->srcpath == NULL)
        G->srcpath = strndup(str,(size_t) len);
      break;
    case 'f':
      G->freq[0] = (float) r4[0]; G->freq[1] = (float) r4[1]; G->freq[2] = (float) r4[2]; G->freq[3] = (float) r4[3];
      break;
    case 'S':
      if (G->nscaff > 0)
        { G->scaffolds[G->nscaff-1].ectg = G->ncontig;
          G->scaffolds[G->nscaff-1].slen = X->spos;
        }
      X->spos = 0;
      if (G->nscaff >= X->sctop)
        { X->sctop = (int) (1.2*G->nscaff) + 500;
          G->scaffolds = realloc(G->scaffolds,sizeof(fga_scaffold)*X->sctop);
        }
      if (G->hdrtot + len + 1 > X->hdrcap)
        { X->hdrcap = (int64_t) (1.2*(G->hdrtot+len+1)) + 10000;
          G->headers = realloc(G->headers,X->hdrcap);
        }
      if (G->scaffolds == NULL || G->headers == NULL)
        { fga_set_error("out of memory");
          return 1;
        }
      G->scaffolds[G->nscaff].hoff = G->hdrtot;
      G->scaffolds[G->nscaff].fctg = G->ncontig;
      memcpy(G->headers+G->hdrtot,str,(size_t) len);
      G->hdrtot += len;
      G->headers[G->hdrtot++] = '\0';
      G->nscaff += 1;
      break;
    case 'G':
      X->spos += ival;
      break;
    case 'C':
      if (G->nscaff == 0)
        { fga_set_error("GDB skeleton: C line before any S line");
          return 1;
        }
      if (G->ncontig >= X->cttop)
        { X->cttop = (int) (1.2*G->ncontig) + 1000;
          G->contigs = realloc(G->contigs,sizeof(fga_contig)*(X->cttop+1));
          if (G->contigs == NULL)
            { fga_set_error("out of memory");
              return 1;
            }
        }
      G->contigs[G->ncontig].boff = X->boff;
      G->contigs[G->ncontig].sbeg = X->spos;
      G->contigs[G->ncontig].clen = ival;
      G->contigs[G->ncontig].scaf = G->nscaff-1;
      G->ncontig += 1;
      if (ival > G->maxctg) G->maxctg = ival;
      G->seqtot += ival;
      X->boff += (ival+3) >> 2;
      X->spos += ival;
      break;
    default:
      break;
  }
  return 0;
}

/* "<len> <text>" of an ASCII ONEcode string field */
static int ascii_string(const char *p, const char **str, int64_t *len)
{ int l = 0, used = 0;
  if (sscanf(p," %d %n",&l,&used) < 1 || l < 0 || (int) strlen(p+used) < l)
    return 1;
  *str = p+used; *len = l;
  return 0;
}

/* ---- binary ONEcode (the container ONElib writes; see fga_aln.c for the layout) ---- */
enum { F_INT = 1, F_REAL, F_CHAR, F_STRING, F_INT_LIST, F_REAL_LIST, F_STRING_LIST, F_DNA };

/* packed integers, list codes and the footer scan live in fga_one.c */

static int read_binary_skeleton(skel_ctx *X, const uint8_t *buf, size_t size, const char *spath)
{ /* field lists per line type from the '~' schema lines of the header */
  uint8_t nfld[128], fld[128][8];
  const uint8_t *p = buf, *end = buf + size;
  int seen_dollar = 0, rc = 1, ci;
  fga_one_codec *codec = calloc(128,sizeof(fga_one_codec));
  uint8_t *dec = NULL;
  int64_t deccap = 0;
  if (codec == NULL) { fga_set_error("out of memory"); return 1; }
  memset(nfld,0xff,sizeof(nfld));

  while (p < end && !seen_dollar)                 /* ASCII header */
    { const uint8_t *e = memchr(p,'\n',(size_t) (end-p));
      size_t n = e ? (size_t) (e-p) : (size_t) (end-p);
      char *ln = strndup((const char *) p,n);
      if (ln == NULL) { fga_set_error("out of memory"); goto done; }
      if (ln[0] == '~' && (ln[2] == 'O' || ln[2] == 'D') && n > 6)
        { int t = (unsigned char) ln[4], k = 0, nf = 0, used = 0;
          const char *q = ln+5;
          if (sscanf(q," %d%n",&nf,&used) == 1 && nf <= 8 && t < 128)
            { q += used;
              for (k = 0; k < nf; k++)
                { int l = 0; char name[32];
                  if (sscanf(q," %d %31s%n",&l,name,&used) < 2) break;
                  q += used;
                  fld[t][k] = !strcmp(name,"INT") ? F_INT : !strcmp(name,"REAL") ? F_REAL : !strcmp(name,"CHAR") ? F_CHAR :
                              !strcmp(name,"STRING") ? F_STRING : !strcmp(name,"INT_LIST") ? F_INT_LIST :
                              !strcmp(name,"REAL_LIST") ? F_REAL_LIST : !strcmp(name,"DNA") ? F_DNA : F_STRING_LIST;
                }
              if (k == nf) nfld[t] = (uint8_t) nf;
            }
        }
      else if (ln[0] == '<')
        { const char *str; int64_t len;
          if (ascii_string(ln+1,&str,&len) == 0 && skel_line(X,'<',str,len,0,NULL)) { free(ln); goto done; }
        }
      else if (ln[0] == '$')
        { if (atoi(ln+1) != 0)
            { fga_set_error("%s is a big-endian ONEcode file",spath);
              free(ln);
              goto done;
            }
          seen_dollar = 1;
        }
      free(ln);
      p += n + (e ? 1 : 0);
    }
  if (!seen_dollar)
    { fga_set_error("%s: binary ONEcode header has no $ line",spath);
      goto done;
    }
  if (fga_one_footer_codecs(buf,size,codec))
    { fga_set_error("%s: malformed ONEcode footer",spath);
      goto done;
    }

  while (p < end && *p != '\n')                   /* data lines until the end-of-data marker */
    { uint8_t x = *p++;
      int k = (x & 0x7f) >> 1, t, i;
      const char *str = NULL;
      int64_t slen = 0, ival = 0;
      double r4[4] = {0,0,0,0};
      int64_t iv[3] = {0,0,0};
      int nr = 0, first_int = 1, ni = 0;
      if (!(x & 0x80) || k >= 52)
        { fga_set_error("%s: unexpected byte 0x%02x in the binary data section",spath,x);
          goto done;
        }
      t = k < 26 ? 'A'+k : 'a'+(k-26);
      if (nfld[t] == 0xff)
        { fga_set_error("%s: line type %c is not in the file's schema",spath,t);
          goto done;
        }
      for (i = 0; i < nfld[t]; i++)
        { int64_t v;
          int u;
          switch (fld[t][i])
          { case F_INT:
              if ((u = fga_one_int(p,end,&v)) == 0) goto trunc;
              p += u;
              if (first_int) { ival = v; first_int = 0; }
              if (ni < 3) iv[ni++] = v;
              break;
            case F_REAL:
              if (p+8 > end) goto trunc;
              if (nr < 4) memcpy(r4+nr,p,8);
              nr += 1; p += 8;
              break;
            case F_CHAR:
              if (p+1 > end) goto trunc;
              p += 1;
              break;
            default:                              /* a list: its length first */
              if ((u = fga_one_int(p,end,&v)) == 0 || v < 0) goto trunc;
              p += u;
              if (fld[t][i] == F_INT_LIST)
                { int64_t f0;
                  if (v > 0)
                    { if ((u = fga_one_int(p,end,&f0)) == 0) goto trunc;
                      p += u;
                      if (v > 1)
                        { int w;
                          if (p >= end) goto trunc;
                          w = *p++;
                          if (x & 1)                  /* compressed differences: not needed, skipped */
                            { int64_t nb;
                              if ((u = fga_one_int(p,end,&nb)) == 0 || nb < 0 || p+u+((nb+7)>>3) > end) goto trunc;
                              p += u + ((nb+7) >> 3);
                            }
                          else
                            { if (w < 1 || w > 8 || p + (v-1)*w > end) goto trunc;
                              p += (v-1)*w;
                            }
                        }
                    }
                }
              else if ((x & 1) && v > 0)              /* a codec-compressed list: bit count, then the code stream */
                { int64_t nb, got;
                  if ((u = fga_one_int(p,end,&nb)) == 0 || nb < 0 || p+u+((nb+7)>>3) > end) goto trunc;
                  p += u;
                  if (fld[t][i] != F_STRING)
                    { p += (nb+7) >> 3;               /* only the scaffold names are read from a GDB */
                      break;
                    }
                  if (!codec[t].have)
                    { fga_set_error("%s: compressed %c lines but no code for them in the footer",spath,t);
                      goto done;
                    }
                  if (v+8 > deccap)
                    { deccap = 2*v + 256;
                      free(dec);
                      dec = malloc((size_t) deccap);
                      if (dec == NULL) { fga_set_error("out of memory"); goto done; }
                    }
                  got = fga_one_codec_decode(codec+t,p,nb,dec,deccap);
                  if (got != v)
                    { fga_set_error("%s: a compressed %c line does not decode to its %lld bytes",spath,t,(long long) v);
                      goto done;
                    }
                  p += (nb+7) >> 3;
                  str = (const char *) dec; slen = v;
                }
              else if (fld[t][i] == F_STRING)
                { if (p+v > end) goto trunc;
                  str = (const char *) p; slen = v; p += v;
                }
              else if (fld[t][i] == F_REAL_LIST)
                { if (p + 8*v > end) goto trunc;
                  p += 8*v;
                }
              else if (fld[t][i] == F_DNA)
                { if (p + ((v+3)>>2) > end) goto trunc;
                  p += (v+3)>>2;
                }
              else
                { fga_set_error("%s: string-list lines are not supported",spath);
                  goto done;
                }
              break;
          }
        }
      if (t == 'M' && X->want_m)
        { if (ni < 3 || skel_mline(X,iv[0],iv[1],iv[2]))
            { if (ni < 3) fga_set_error("%s: M line with fewer than three integers",spath);
              goto done;
            }
          continue;
        }
      if (skel_line(X,(char) t,str,slen,ival,r4))
        goto done;
    }
  rc = 0;
  goto done;

trunc:
  fga_set_error("%s: truncated or malformed binary ONEcode line",spath);
done:
  for (ci = 0; ci < 128; ci++)
    free(codec[ci].look);
  free(codec); free(dec);
  return rc;
}

int fga_gdb_open(const char *path, fga_gdb **out)
{ fga_gdb *G;
  char *noext, *dir, *root, *spath = NULL, *bpath = NULL;
  FILE *f;
  char *line = NULL;
  size_t cap = 0;
  ssize_t n;
  skel_ctx X;
  int first = 1;
  int64_t boff;

  *out = NULL;
  G = calloc(1,sizeof(fga_gdb));
  if (G == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  memset(&X,0,sizeof(X));
  X.G = G;

  noext = strip_gdb_ext(path);
  dir   = fga_path_dir(noext);
  root  = fga_path_root(noext,NULL);

  /* <root>.1gdb (binary or ASCII ONEcode, what the reference writes) first, then <root>.gdb */
  if (asprintf(&spath,"%s/%s.1gdb",dir,root) < 0) spath = NULL;
  f = (spath != NULL) ? fopen(spath,"r") : NULL;
  if (f == NULL)
    { free(spath);
      if (asprintf(&spath,"%s/%s.gdb",dir,root) < 0) spath = NULL;
      f = (spath != NULL) ? fopen(spath,"r") : NULL;
    }
  if (f == NULL)
    { fga_set_error("cannot find/open GDB skeleton %s/%s.1gdb or .gdb",dir,root);
      goto fail;
    }

  { /* binary container?  The header of a binary file ends with a "$ <endian>" line */
    uint8_t *buf = NULL;
    size_t   size = 0, got;
    int      binary = 0;
    if (fseek(f,0,SEEK_END) == 0)
      { long sz = ftell(f);
        rewind(f);
        if (sz > 0 && (buf = malloc((size_t) sz + 1)) != NULL)
          { got = fread(buf,1,(size_t) sz,f);
            size = got;
            buf[size] = '\0';
            { const uint8_t *q = buf;
              while (q < buf+size)                  /* scan the ASCII header lines only */
                { const uint8_t *e = memchr(q,'\n',(size_t) (buf+size-q));
                  if (q[0] == '$') { binary = 1; break; }
                  if (e == NULL || (q[0] & 0x80) || (q[0] >= 'A' && q[0] <= 'z' && q[0] != '~'))
                    break;
                  q = e+1;
                }
            }
          }
        rewind(f);
      }
    if (binary)
      { int rc;
        if (size < 8 || memcmp(buf,"1 ",2) != 0 || strstr((char *) buf," gdb ") == NULL)
          { fga_set_error("%s is not a ONEcode gdb file",spath);
            free(buf);
            goto fail_f;
          }
        rc = read_binary_skeleton(&X,buf,size,spath);
        free(buf);
        if (rc) goto fail_f;
        goto parsed;
      }
    free(buf);
  }

  while ((n = getline(&line,&cap,f)) > 0)
    { while (n > 0 && (line[n-1] == '\n' || line[n-1] == '\r'))
        line[--n] = '\0';
      if (first)
        { first = 0;
          if (line[0] != '1' || strstr(line," gdb ") == NULL)
            { fga_set_error("%s is not a ONEcode gdb file",spath);
              goto fail_f;
            }
          continue;
        }
      switch (line[0])
      { case '<': case 'S':
          { const char *str; int64_t len;
            if (ascii_string(line+1,&str,&len))
              { if (line[0] == '<') break;
                fga_set_error("%s: malformed S line",spath);
                goto fail_f;
              }
            if (skel_line(&X,line[0],str,len,0,NULL)) goto fail_f;
          }
          break;
        case 'f':
          { double r4[4];
            if (sscanf(line+1," %lf %lf %lf %lf",r4,r4+1,r4+2,r4+3) != 4)
              { fga_set_error("%s: malformed f line",spath);
                goto fail_f;
              }
            if (skel_line(&X,'f',NULL,0,0,r4)) goto fail_f;
          }
          break;
        case 'G': case 'C':
          if (skel_line(&X,line[0],NULL,0,atoll(line+1),NULL)) goto fail_f;
          break;
        default:     /* header / provenance / schema / count lines, deprecated u and M lines */
          break;
      }
    }
parsed:
  fclose(f);
  f = NULL;
  boff = X.boff;
  if (G->nscaff == 0 || G->ncontig == 0)
    { fga_set_error("%s holds no scaffolds/contigs",spath);
      goto fail;
    }
  G->scaffolds[G->nscaff-1].ectg = G->ncontig;
  G->scaffolds[G->nscaff-1].slen = X.spos;
  if (G->srcpath == NULL)
    G->srcpath = strdup("");
  G->path = strdup(spath);

  if (asprintf(&bpath,"%s/.%s.bps",dir,root) < 0)
    { fga_set_error("out of memory");
      goto fail;
    }
  { FILE *b = fopen(bpath,"r");
    if (b == NULL)
      { fga_set_error("cannot open .bps file %s for GDB %s",bpath,path);
        goto fail;
      }
    G->bpslen = boff;
    G->bps = malloc(boff+16);
    if (G->bps == NULL)
      { fga_set_error("out of memory loading %s",bpath);
        fclose(b);
        goto fail;
      }
    if (boff > 0 && read_image(b,G->bps,boff))
      { fga_set_error("%s is shorter than its skeleton says (%lld bytes)",bpath,(long long) boff);
        fclose(b);
        goto fail;
      }
    memset(G->bps+boff,0,16);
    fclose(b);
  }

  { char *mpath = NULL;
    FILE *mf;
    if (asprintf(&mpath,"%s/.%s.msk",dir,root) >= 0 && (mf = fopen(mpath,"r")) != NULL)
      { int64_t hdr[2], *ctg = NULL, i;
        if (fread(hdr,sizeof(int64_t),2,mf) == 2 && hdr[0] == G->ncontig && hdr[1] > 0)
          { G->nmask = hdr[1];
            ctg = malloc(sizeof(int64_t)*G->nmask);
            G->mbeg = malloc(sizeof(int64_t)*G->nmask);
            G->mend = malloc(sizeof(int64_t)*G->nmask);
            G->moff = calloc(G->ncontig+1,sizeof(int64_t));
            if (ctg != NULL && G->mbeg != NULL && G->mend != NULL && G->moff != NULL &&
                fread(ctg,sizeof(int64_t),G->nmask,mf) == (size_t) G->nmask &&
                fread(G->mbeg,sizeof(int64_t),G->nmask,mf) == (size_t) G->nmask &&
                fread(G->mend,sizeof(int64_t),G->nmask,mf) == (size_t) G->nmask)
              { for (i = 0; i < G->nmask; i++)
                  G->moff[ctg[i]+1] += 1;
                for (i = 0; i < G->ncontig; i++)
                  G->moff[i+1] += G->moff[i];
              }
            else
              G->nmask = 0;
            free(ctg);
          }
        fclose(mf);
      }
    free(mpath);
  }

  free(line); free(noext); free(dir); free(root); free(spath); free(bpath);
  *out = G;
  return 0;

fail_f:
  if (f != NULL) fclose(f);
fail:
  free(line); free(noext); free(dir); free(root); free(spath); free(bpath);
  free(G->scaffolds); free(G->contigs); free(G->headers); free(G->srcpath); free(G->path);
  free(G->bps);
  free(G);
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 *  Mask files: `#<mask>[.1ano]` arguments (FastGA.c:4568-4573 -> GIXmake.c:1821-1842)
 *
 *  A .1ano / .ano file (ONEcode, binary or ASCII; Read_ANO, ANO.c:105-512) holds a GDB skeleton and M lines
 *  (scaffold index, beg, end) in scaffold coordinates.  Its scaffolds are matched to the GDB's by name, length and
 *  contig structure (Are_Skeletons_Equal, GDB.c:2094-2132), every interval goes to the contig its start lies in
 *  (ANO.c:445-486), and the masks named on the command line are united per contig: intervals in order of their start,
 *  one that starts AFTER the end so far begins a new interval (ANO_Union, ANO.c:747-757) -- the soft mask the index
 *  build reads (setup_thread_with_masks, GIXmake.c:1100-1108; fga_gix.c / fga_gixdev.hip).
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int64_t ctg, beg, end; } mask_iv;

static int mask_iv_cmp(const void *l, const void *r)
{ const mask_iv *a = l, *b = r;
  if (a->ctg != b->ctg) return a->ctg < b->ctg ? -1 : 1;
  if (a->beg != b->beg) return a->beg < b->beg ? -1 : 1;
  return 0;
}

/* the intervals of one mask file, in contig coordinates of G, appended to *iv */
static int ano_read(const fga_gdb *G, const char *path, mask_iv **iv, int64_t *niv, int64_t *cap)
{ char *root = NULL, *fname = NULL;
  size_t n = strlen(path);
  uint8_t *buf = NULL;
  size_t size = 0;
  FILE *f = NULL;
  fga_gdb *A = NULL;
  skel_ctx X;
  int *map = NULL;
  int rc = 1, binary = 0, s, t, c;
  int64_t i;

  memset(&X,0,sizeof(X));
  if (n > 5 && strcmp(path+n-5,".1ano") == 0) n -= 5;
  else if (n > 4 && strcmp(path+n-4,".ano") == 0) n -= 4;
  if (asprintf(&root,"%.*s",(int) n,path) < 0) { root = NULL; goto oom; }
  if (asprintf(&fname,"%s.1ano",root) < 0) { fname = NULL; goto oom; }
  f = fopen(fname,"r");
  if (f == NULL)
    { free(fname);
      if (asprintf(&fname,"%s.ano",root) < 0) { fname = NULL; goto oom; }
      f = fopen(fname,"r");
    }
  if (f == NULL)
    { fga_set_error("Cannot find/open ANO file %s",path);
      goto done;
    }
  if (fseek(f,0,SEEK_END) == 0)
    { long sz = ftell(f);
      rewind(f);
      if (sz > 0 && (buf = malloc((size_t) sz + 1)) != NULL)
        { size = fread(buf,1,(size_t) sz,f);
          buf[size] = '\0';
        }
    }
  fclose(f); f = NULL;
  if (buf == NULL || size < 8 || memcmp(buf,"1 ",2) != 0 || strstr((char *) buf," ano ") == NULL)
    { fga_set_error("%s is not a ONEcode ano file",fname);
      goto done;
    }
  { const uint8_t *q = buf;                         /* binary?  The header of a binary file ends with a "$ <endian>" line */
    while (q < buf+size)
      { const uint8_t *e = memchr(q,'\n',(size_t) (buf+size-q));
        if (q[0] == '$') { binary = 1; break; }
        if (e == NULL || (q[0] & 0x80) || (q[0] >= 'A' && q[0] <= 'z' && q[0] != '~'))
          break;
        q = e+1;
      }
  }
  A = calloc(1,sizeof(fga_gdb));
  if (A == NULL) goto oom;
  X.G = A; X.want_m = 1;
  if (binary)
    { if (read_binary_skeleton(&X,buf,size,fname)) goto done; }
  else
    { char *ln = (char *) buf;
      while (ln != NULL && *ln != '\0')
        { char *e = strchr(ln,'\n');
          if (e != NULL) *e = '\0';
          if (ln[0] != '\0' && (ln[1] == ' ' || ln[1] == '\0'))
            switch (ln[0])
            { case 'S':
                { const char *str; int64_t len;
                  if (ascii_string(ln+1,&str,&len))
                    { fga_set_error("%s: malformed S line",fname); goto done; }
                  if (skel_line(&X,'S',str,len,0,NULL)) goto done;
                }
                break;
              case 'G': case 'C':
                if (skel_line(&X,ln[0],NULL,0,atoll(ln+1),NULL)) goto done;
                break;
              case 'M':
                { long long a, b, cc;
                  if (sscanf(ln+1," %lld %lld %lld",&a,&b,&cc) != 3)
                    { fga_set_error("%s: malformed M line",fname); goto done; }
                  if (skel_mline(&X,a,b,cc)) goto done;
                }
                break;
              default:
                break;
            }
          ln = e != NULL ? e+1 : NULL;
        }
    }
  if (A->nscaff == 0)
    { fga_set_error(".1ano does not contain prefacing GDB skeleton");
      goto done;
    }
  A->scaffolds[A->nscaff-1].ectg = A->ncontig;
  A->scaffolds[A->nscaff-1].slen = X.spos;

  /* its scaffolds in terms of the GDB's */
  if (A->nscaff != G->nscaff)
    { fga_set_error("%s: GDB structures not equivalent",fname);
      goto done;
    }
  map = malloc(sizeof(int)*(A->nscaff > 0 ? A->nscaff : 1));
  if (map == NULL) goto oom;
  { char *used = calloc(G->nscaff > 0 ? G->nscaff : 1,1);
    if (used == NULL) goto oom;
    for (s = 0; s < A->nscaff; s++)
      { const fga_scaffold *sa = A->scaffolds + s;
        map[s] = -1;
        for (t = 0; t < G->nscaff; t++)
          { const fga_scaffold *ta = G->scaffolds + t;
            if (used[t] || ta->slen != sa->slen || strcmp(G->headers+ta->hoff,A->headers+sa->hoff) != 0 ||
                ta->ectg - ta->fctg != sa->ectg - sa->fctg)
              continue;
            for (c = ta->fctg; c < ta->ectg; c++)
              if (G->contigs[c].clen != A->contigs[sa->fctg+(c-ta->fctg)].clen ||
                  G->contigs[c].sbeg != A->contigs[sa->fctg+(c-ta->fctg)].sbeg)
                break;
            if (c < ta->ectg)
              continue;
            map[s] = t; used[t] = 1;
            break;
          }
        if (map[s] < 0)
          { free(used);
            fga_set_error("%s: GDB structures not equivalent",fname);
            goto done;
          }
      }
    free(used);
  }
  for (i = 0; i < X.nm; i++)
    { int64_t scf = X.mrec[3*i], beg = X.mrec[3*i+1], end = X.mrec[3*i+2];
      const fga_scaffold *ts;
      if (scf < 0 || scf >= A->nscaff)
        { fga_set_error("%s: %lld'th scaffold not declared in prolog",fname,(long long) scf+1);
          goto done;
        }
      if (beg > end) { const int64_t x = beg; beg = end; end = x; }
      ts = G->scaffolds + map[scf];
      c = ts->fctg;
      while (c+1 < ts->ectg && beg >= G->contigs[c+1].sbeg)
        c += 1;
      if (*niv >= *cap)
        { mask_iv *nv;
          *cap = *cap ? 2*(*cap) : 4096;
          nv = realloc(*iv,sizeof(mask_iv)*(*cap));
          if (nv == NULL) goto oom;
          *iv = nv;
        }
      (*iv)[*niv].ctg = c; (*iv)[*niv].beg = beg - G->contigs[c].sbeg; (*iv)[*niv].end = end - G->contigs[c].sbeg;
      *niv += 1;
    }
  rc = 0;
  goto done;
oom:
  fga_set_error("out of memory");
done:
  if (f != NULL) fclose(f);
  if (A != NULL) { free(A->scaffolds); free(A->contigs); free(A->headers); free(A->srcpath); free(A); }
  free(X.mrec); free(map); free(buf); free(root); free(fname);
  return rc;
}

/* the soft mask of G becomes the union of the masks named: paths[k] = a .1ano / .ano file, or "" for the GDB's own mask
   (the lower-case runs of its FASTA: what a bare `#` names, GIXmake.c:1829-1832) */
int fga_gdb_apply_masks(fga_gdb *G, const char *const *paths, int npaths)
{ mask_iv *iv = NULL;
  int64_t niv = 0, cap = 0, i, k, out = 0, *moff = NULL, *mbeg = NULL, *mend = NULL;
  int p;
  if (G == NULL || (npaths > 0 && paths == NULL))
    { fga_set_error("fga_gdb_apply_masks: null argument");
      return 1;
    }
  for (p = 0; p < npaths; p++)
    if ((paths[p] == NULL || paths[p][0] == '\0') && G->nmask == 0 && G->path != NULL)
      { /* the GDB's own mask, of a GDB the reference's FAtoGDB made: <root>.1ano beside the skeleton (GIXmake.c:1829-1832) */
        char *ap = NULL;
        size_t n = strlen(G->path);
        FILE *tf;
        if (n > 5 && strcmp(G->path+n-5,".1gdb") == 0) n -= 5;
        else if (n > 4 && strcmp(G->path+n-4,".gdb") == 0) n -= 4;
        if (asprintf(&ap,"%.*s.1ano",(int) n,G->path) < 0) { free(iv); fga_set_error("out of memory"); return 1; }
        if ((tf = fopen(ap,"r")) != NULL)
          { fclose(tf);
            if (ano_read(G,ap,&iv,&niv,&cap)) { free(ap); free(iv); return 1; }
          }
        free(ap);
      }
    else if (paths[p] == NULL || paths[p][0] == '\0')
      { int c;
        for (c = 0; c < G->ncontig && G->nmask > 0; c++)
          for (i = G->moff[c]; i < G->moff[c+1]; i++)
            { if (niv >= cap)
                { mask_iv *nv;
                  cap = cap ? 2*cap : 4096;
                  nv = realloc(iv,sizeof(mask_iv)*cap);
                  if (nv == NULL) { free(iv); fga_set_error("out of memory"); return 1; }
                  iv = nv;
                }
              iv[niv].ctg = c; iv[niv].beg = G->mbeg[i]; iv[niv].end = G->mend[i];
              niv += 1;
            }
      }
    else if (ano_read(G,paths[p],&iv,&niv,&cap))
      { free(iv);
        return 1;
      }
  if (niv > 1)
    qsort(iv,niv,sizeof(mask_iv),mask_iv_cmp);
  moff = calloc(G->ncontig+1,sizeof(int64_t));
  mbeg = malloc(sizeof(int64_t)*(niv > 0 ? niv : 1));
  mend = malloc(sizeof(int64_t)*(niv > 0 ? niv : 1));
  if (moff == NULL || mbeg == NULL || mend == NULL)
    { free(iv); free(moff); free(mbeg); free(mend);
      fga_set_error("out of memory");
      return 1;
    }
  for (i = 0; i < niv; i = k)
    { int64_t end = -1;
      for (k = i; k < niv && iv[k].ctg == iv[i].ctg; k++)
        if (iv[k].beg > end)
          { mbeg[out] = iv[k].beg; mend[out] = end = iv[k].end;
            out += 1;
            moff[iv[i].ctg+1] += 1;
          }
        else if (iv[k].end > end)
          mend[out-1] = end = iv[k].end;
    }
  for (p = 0; p < G->ncontig; p++)
    moff[p+1] += moff[p];
  free(G->moff); free(G->mbeg); free(G->mend);
  G->moff = moff; G->mbeg = mbeg; G->mend = mend; G->nmask = out;
  free(iv);
  return 0;
}

void fga_gdb_close(fga_gdb *G)
{ if (G == NULL) return;
  free(G->scaffolds); free(G->contigs); free(G->headers); free(G->srcpath); free(G->path);
  free(G->bps);
  free(G->moff); free(G->mbeg); free(G->mend);
  free(G);
}

int     fga_gdb_ncontig(const fga_gdb *G)            { return G->ncontig; }
int     fga_gdb_nscaff(const fga_gdb *G)             { return G->nscaff; }
int64_t fga_gdb_seqtot(const fga_gdb *G)             { return G->seqtot; }
int64_t fga_gdb_maxctg(const fga_gdb *G)             { return G->maxctg; }
int64_t fga_gdb_contig_len(const fga_gdb *G, int c)  { return G->contigs[c].clen; }
int64_t fga_gdb_nmask(const fga_gdb *G)              { return G->nmask; }
void    fga_gdb_freq(const fga_gdb *G, float *f4)    { memcpy(f4,G->freq,4*sizeof(float)); }

/* Unpack contig c into numeric form (0..3), buf must hold clen+2 bytes; buf[0] and buf[clen+1] get the
 * sentinel 4 the aligner needs either side (GDB.c:1718-1727, gene_core.c:397); returns buf+1. */
uint8_t *fga_gdb_get_contig(const fga_gdb *G, int c, uint8_t *buf)
{ static uint32_t quad[256];            /* the four bases of a packed byte as four numeric bytes */
  static volatile int quad_ready = 0;
  const uint8_t *src = G->bps + G->contigs[c].boff;
  int64_t len = G->contigs[c].clen, i;
  uint8_t *s = buf+1;
  if (!quad_ready)                      /* every writer stores the same values, so a race here is harmless */
    { int b, q;
      for (b = 0; b < 256; b++)
        { uint8_t f[4];
          for (q = 0; q < 4; q++)
            f[q] = (uint8_t) ((b >> (2*q)) & 3);
          memcpy(quad+b,f,4);
        }
      quad_ready = 1;
    }
  buf[0] = 4;
  for (i = 0; i+4 <= len; i += 4)
    memcpy(s+i,quad + src[i>>2],4);
  for (; i < len; i++)
    s[i] = (src[i>>2] >> (2*(i&3))) & 3;
  s[len] = 4;
  return s;
}

int fga_gdb_write_skeleton(const fga_gdb *G, const char *path, const char *prog, const char *command)
{ return write_skeleton_file(G,path,prog,command); }
