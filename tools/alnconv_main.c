/* alnconv_main.c -- ALNtoPAF / ALNtoPSL command lines on top of libfastga_amd (the PSL tool is this file with -DALN_PSL).
 *
 *   ALNtoPAF [-mxsSw] [-T<int(8)>] <alignment:path>[.1aln]         (reference grammar: ALNtoPAF.c:28, 662-704)
 *   ALNtoPSL [-T<int(8)>] <alignment:path>[.1aln]                  (ALNtoPSL.c)
 *
 * The .1aln is read in-process (fga_read_1aln: the reference's own files, list codecs included, or ours), the two GDBs
 * are found through the file's reference lines like the reference does, base-level output (-m -x -s -S, PSL) gets its
 * edit scripts from the device (fga_trace_pts) and the text is written to stdout.  Plain PAF needs no GPU.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fastga_amd.h"

#ifdef ALN_PSL
#define PROG "ALNtoPSL"
#else
#define PROG "ALNtoPAF"
#endif

int main(int argc, char *argv[])
{ const char *src = NULL;
  char *path = NULL, *db1 = NULL, *db2 = NULL;
  fga_alns *alns = NULL;
  fga_gdb *g1 = NULL, *g2 = NULL;
  fga_dev *dev = NULL;
  fga_dgenome *d1 = NULL, *d2 = NULL;
  fga_traces *tr = NULL;
  int flags = 0, nthreads = 8, tspace = 100, i, rc = 1, bases;

  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      { const char *f;
        if (argv[i][1] == 'T')
          { nthreads = atoi(argv[i]+2);
            if (nthreads < 1)
              { fprintf(stderr,"%s: Number of threads must be positive\n",PROG);
                return 1;
              }
            continue;
          }
        for (f = argv[i]+1; *f; f++)
          switch (*f)
          {
#ifndef ALN_PSL
            case 'm': flags |= FGA_PAF_CIGAR_M; break;
            case 'x': flags |= FGA_PAF_CIGAR_X; break;
            case 's': flags |= FGA_PAF_CS_SHORT; break;
            case 'S': flags |= FGA_PAF_CS_LONG; break;
            case 'w': flags |= FGA_PAF_SWAP; break;
#endif
            default:
              fprintf(stderr,"%s: -%c is an illegal option\n",PROG,*f);
              return 1;
          }
      }
    else if (src == NULL)
      src = argv[i];
  if (src == NULL)
    {
#ifdef ALN_PSL
      fprintf(stderr,"Usage: %s [-T<int(8)>] <alignment:path>[.1aln]\n",PROG);
#else
      fprintf(stderr,"Usage: %s [-mxsSw] [-T<int(8)>] <alignment:path>[.1aln]\n",PROG);
#endif
      return 1;
    }
  if ((flags & FGA_PAF_CIGAR_M) && (flags & FGA_PAF_CIGAR_X))
    { fprintf(stderr,"%s: Only one of -m or -x can be set\n",PROG);
      return 1;
    }
  if ((flags & FGA_PAF_CS_SHORT) && (flags & FGA_PAF_CS_LONG))
    { fprintf(stderr,"%s: Only one of -s or -S can be set\n",PROG);
      return 1;
    }
  { size_t n = strlen(src);
    if (n > 5 && strcmp(src+n-5,".1aln") == 0)
      path = strdup(src);
    else if (asprintf(&path,"%s.1aln",src) < 0)
      path = NULL;
    if (path == NULL) return 1;
  }
#ifdef ALN_PSL
  bases = 1;
#else
  bases = (flags & (FGA_PAF_CIGAR_M|FGA_PAF_CIGAR_X|FGA_PAF_CS_SHORT|FGA_PAF_CS_LONG)) != 0;
#endif

  if (fga_read_1aln(path,&alns,&tspace,&db1,&db2)) goto fail;
  if (db1 == NULL)
    { fprintf(stderr,"%s: %s does not name its genome data bases\n",PROG,path);
      goto done;
    }
  if (fga_gdb_open(db1,&g1)) goto fail;
  if (db2 != NULL && fga_gdb_open(db2,&g2)) goto fail;
  if (bases)
    { if (fga_dev_open(0,&dev)) goto fail;
      fga_dev_set_host_threads(dev,nthreads);
      if (fga_dgenome_upload(dev,g1,NULL,0,g2 == NULL,&d1)) goto fail;
      if (g2 != NULL && fga_dgenome_upload(dev,g2,NULL,0,1,&d2)) goto fail;
      if (fga_trace_pts_regrouped(dev,d1,g2 != NULL ? d2 : d1,alns,tspace,0,&tr)) goto fail;
    }
#ifdef ALN_PSL
  if (fga_write_psl("-",g1,g2,alns,tr,nthreads)) goto fail;
#else
  if (fga_write_paf("-",g1,g2,alns,tr,flags,nthreads)) goto fail;
#endif
  rc = 0;
  goto done;

fail:
  fprintf(stderr,"%s: %s\n",PROG,fga_last_error());
done:
  fga_traces_free(tr);
  fga_dgenome_free(d2); fga_dgenome_free(d1);
  if (dev) fga_dev_close(dev);
  if (g2) fga_gdb_close(g2);
  if (g1) fga_gdb_close(g1);
  fga_alns_free(alns);
  free(db1); free(db2); free(path);
  return rc;
}
