/* FastGA -- drop-in command line for the MI355X hot path (links libfastga_amd.so).
 *
 * Accepts the reference grammar (FastGA.c:62-66, README "FastGA"):
 *   FastGA [-vkMS] [-L:<log>] [-T<int(8)>] [-P<dir>] [-1:<out>] [-f<int(10)>] [-c<int(85)>] [-s<int(1000)>]
 *          [-l<int(100)>] [-i<float(.7)>]  <source1>[.gdb|.1gdb|.gix|.fa...]  [<source2>]
 * What differs: the sub-process glue is in-process -- a missing GDB / GIX is built with this library's own
 * producers (fga_fasta_to_gdb, fga_gix_build) instead of system("FAtoGDB"/"GIXmake"), and only the .1aln output
 * (-1:<name>) PAF (-paf[mxsS]*, the default) and PSL (-psl), both on stdout, are produced natively.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "fastga_amd.h"

static int exists(const char *fmt, const char *root)
{ char *p;
  int r;
  if (asprintf(&p,fmt,root) < 0) return 0;
  r = access(p,R_OK) == 0;
  free(p);
  return r;
}

/* strip a known extension; returns malloc'd root */
static char *root_of(const char *src, int *is_fasta)
{ static const char *ext[] = { ".1gdb", ".gdb", ".gix", ".fa.gz", ".fna.gz", ".fasta.gz", ".fa", ".fna", ".fasta", NULL };
  char *r = strdup(src);
  size_t n = strlen(r);
  int i;
  *is_fasta = 0;
  for (i = 0; ext[i] != NULL; i++)
    { size_t m = strlen(ext[i]);
      if (n > m && strcmp(r+n-m,ext[i]) == 0)
        { r[n-m] = '\0';
          *is_fasta = (i >= 3);
          break;
        }
    }
  return r;
}

static int prepare(const char *src, char **root, int nthreads, int verbose, int want_gix_files)
{ int isfa;
  char *r = root_of(src,&isfa);
  *root = r;
  if (!exists("%s.1gdb",r) && !exists("%s.gdb",r))          /* both ONEcode forms of the skeleton are read */
    { static const char *fext[] = { ".fa", ".fna", ".fasta", ".fa.gz", ".fna.gz", ".fasta.gz", NULL };
      char *fa = NULL;
      int i;
      if (isfa)
        fa = strdup(src);
      else
        for (i = 0; fext[i] != NULL && fa == NULL; i++)
          { char *p;
            if (asprintf(&p,"%s%s",r,fext[i]) >= 0)
              { if (access(p,R_OK) == 0) fa = p; else free(p); }
          }
      if (fa == NULL)
        { fprintf(stderr,"FastGA: cannot find a GDB or FASTA for %s\n",src);
          return 1;
        }
      if (verbose) fprintf(stderr,"  Creating genome data base (GDB) %s.gdb\n",r);
      if (fga_fasta_to_gdb(fa,r,0))
        { fprintf(stderr,"FastGA: %s\n",fga_last_error());
          return 1;
        }
      free(fa);
    }
  if (!exists("%s.gix",r) && want_gix_files)        /* otherwise the index is built on the device, in HBM only */
    { fga_gdb *g;
      if (verbose) fprintf(stderr,"  Creating genome index (GIX) %s.gix\n",r);
      if (fga_gdb_open(r,&g) || fga_gix_build(g,r,nthreads))
        { fprintf(stderr,"FastGA: %s\n",fga_last_error());
          return 1;
        }
      fga_gdb_close(g);
    }
  return 0;
}

int main(int argc, char *argv[])
{ fga_run_params P;
  fga_run_stats S;
  char *src[2] = { NULL, NULL }, *root[2] = { NULL, NULL }, *out = NULL, *outpath = NULL;
  int nsrc = 0, verbose = 0, keep = 0, paf = 0, i;
  int cmin = 85, cbreak = 1000;
  double ident = .7;
  char cmd[4096];
  size_t cl = 0;

  memset(&P,0,sizeof(P));
  P.freq = 10; P.align_min = 100; P.nthreads = 8;
  cmd[0] = '\0';
  for (i = 0; i < argc && cl < sizeof(cmd)-2; i++)
    cl += snprintf(cmd+cl,sizeof(cmd)-cl,"%s%s",i ? " " : "",argv[i]);

  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      switch (argv[i][1])
      { case '1':
          if (argv[i][2] != ':') { fprintf(stderr,"FastGA: -1 must be followed by :<name>\n"); return 1; }
          out = argv[i]+3;
          break;
        case 'f': P.freq = atoi(argv[i]+2); break;
        case 'c': cmin = atoi(argv[i]+2); break;
        case 's': cbreak = atoi(argv[i]+2); break;
        case 'l': P.align_min = atoi(argv[i]+2); break;
        case 'i': ident = atof(argv[i]+2); break;
        case 'T': P.nthreads = atoi(argv[i]+2); break;
        case 'P': case 'L': break;
        case 'p':
          if (strncmp(argv[i]+1,"paf",3) == 0)
            { const char *f;
              paf = 1;
              for (f = argv[i]+4; *f; f++)
                switch (*f)
                { case 'm': P.paf_flags |= FGA_PAF_CIGAR_M; break;
                  case 'x': P.paf_flags |= FGA_PAF_CIGAR_X; break;
                  case 's': P.paf_flags |= FGA_PAF_CS_SHORT; break;
                  case 'S': P.paf_flags |= FGA_PAF_CS_LONG; break;
                  default:
                    fprintf(stderr,"FastGA: Just one or more of m, x, s or S can follow -paf\n");
                    return 1;
                }
              break;
            }
          if (strcmp(argv[i]+1,"psl") == 0)
            { paf = 1;
              P.paf_flags = FGA_OUT_PSL;
              break;
            }
          fprintf(stderr,"FastGA: -%s is an illegal option\n",argv[i]+1);
          return 1;
        default:
          { const char *f;
            for (f = argv[i]+1; *f; f++)
              switch (*f)
              { case 'v': verbose = 1; break;
                case 'k': keep = 1; break;
                case 'M': P.soft_mask = 1; break;
                case 'S': P.symmetric = 1; break;
                default:
                  fprintf(stderr,"FastGA: -%c is an illegal option\n",*f);
                  return 1;
              }
          }
      }
    else if (argv[i][0] == '#')
      P.soft_mask = 1;
    else if (nsrc < 2)
      src[nsrc++] = argv[i];
  if (nsrc == 0)
    { fprintf(stderr,"Usage: FastGA [-vkMS] [-T<int(8)>] [-f<int(10)>] [-c<int(85)>] [-s<int(1000)>] [-l<int(100)>]"
                     " [-i<float(.7)>] [-paf[mxsS]* | -psl | -1:<out>] <source1> [<source2>]\n");
      return 1;
    }
  if (out == NULL)                       /* like the reference, PAF on stdout is the default output */
    paf = 1;
  if (paf)
    P.paf_path = "-";
  if (ident < .55 || ident >= 1.)
    { fprintf(stderr,"FastGA: Minimum alignment similarity %g must be in [0.55,1.0)\n",ident);
      return 1;
    }
  P.chain_min = 2*cmin; P.chain_break = 2*cbreak;
  P.align_rate = 1.-ident;
  P.command_line = cmd;
  if (out != NULL)
  { size_t n = strlen(out);
    if (n > 5 && strcmp(out+n-5,".1aln") == 0)
      outpath = strdup(out);
    else if (asprintf(&outpath,"%s.1aln",out) < 0)
      return 1;
    P.out_path = outpath;
  }
  for (i = 0; i < nsrc; i++)
    if (prepare(src[i],root+i,P.nthreads,verbose,keep))
      return 1;
  if (nsrc == 2 && strcmp(root[0],root[1]) == 0)
    nsrc = 1;
  if (verbose) fprintf(stderr,"\n  Using GPU %d and %d host threads\n",P.device,P.nthreads);
  if (fga_run(root[0],nsrc == 2 ? root[1] : NULL,&P,&S))
    { fprintf(stderr,"FastGA: %s\n",fga_last_error());
      return 1;
    }
  if (verbose)
    { fprintf(stderr,"\n  Total seeds = %lld, ave. len = %.1f\n",(long long) S.nseeds,
                     S.nseeds ? (1.*S.seed_len_sum)/S.nseeds : 0.);
      fprintf(stderr,"  Resources for phase:  merge %.3fs (kernel %.3f ms)\n",S.merge_s,S.merge_kernel_ms);
      fprintf(stderr,"\n  Total hits over %dbp = %lld, %lld aln's, %lld non-redundant aln's of ave len %lld\n",
                     cmin,(long long) S.nhits,(long long) S.nalns,(long long) S.nlive,
                     (long long) (S.nlive ? S.cover/S.nlive : 0));
      fprintf(stderr,"  Resources for phase:  sort %.3fs (kernel %.3f ms) download %.3fs chain %.3fs extend %.3fs"
                     " (kernel %.3f ms, %lld calls, %lld waves) filter %.3fs write %.3fs\n",
                     S.sort_s,S.sort_kernel_ms,S.download_s,S.chain_s,S.extend_s,S.extend_kernel_ms,
                     (long long) S.ncalls,(long long) S.nwaves,S.filter_s,S.write_s);
      if (paf)
        fprintf(stderr,"  PAF: edit scripts %.3fs (kernels %.3f ms), regroup + format %.3fs\n",
                       S.trace_s,S.trace_kernel_ms,S.paf_s);
      fprintf(stderr,"  Load %.3fs  upload %.3fs\n",S.load_s,S.upload_s);
    }
  free(outpath);
  return 0;
}
