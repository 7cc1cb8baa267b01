/* fga_aln.c -- .1aln writer (ASCII ONEcode).
 *
 * Replaces open_Aln_Write / Write_Aln_Overlap / Write_Aln_Trace (reference alncode.c:239-305) and Write_Skeleton
 * (GDB.c:2065-2092) as used by la_merge (FastGA.c:4049-4113).  The reference writes the *binary* ONEcode form
 * through its vendored ONElib; this writer emits the equivalent *ASCII* ONEcode file (same schema alncode.c:19-52,
 * same line content and order), which the reference's own readers (ONEview, ALNshow, ONEaln) accept and which
 * `ONEview` prints identically apart from the '!' provenance and '<' path lines.  An ASCII
 * file must embed its schema ('~' lines) and its count lines ('#', '@', '+', '%') ahead of the data, so the
 * statistics are accumulated first and the body is written second.
 *   body:  t <tspace> / g + skeleton of genome 1 / g + skeleton of genome 2 (not for self) /
 *          per alignment: A aread abpos aepos bread bbpos bepos / R (complement) / D diffs /
 *                         T n b-deltas (odd trace bytes) / X n diffs (even trace bytes)
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <pthread.h>

#include "fga_host.h"
#include "fastga_amd.h"

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

  f = fopen(path,"w");
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
    int nth = 8, t;
    fmt_job  job[8];
    pthread_t th[8];
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
