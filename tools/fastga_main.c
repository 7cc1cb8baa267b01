/* FastGA -- drop-in command line for the MI355X hot path (links libfastga_amd.so).
 *
 * Accepts the reference grammar (FastGA.c:62-66, 4444-4637; README "FastGA"):
 *   FastGA [-vkMS] [-L:<log>] [-T<int(8)>] [-P<dir>] [<format(-paf)>] [-f<int(10)>] [-c<int(85)>] [-s<int(1000)>]
 *          [-l<int(100)>] [-i<float(.7)>]  <source1>[.gdb|.1gdb|.gix|.fa...]  [<source2>]
 *          <format> = -paf[mxsS]* | -psl | -1:<align:path>[.1aln]
 * plus one option of its own: -G<n> (or FGA_DEVICES=<id>,...) cuts the comparison over n GPUs of the node (fga_run_multi:
 * the reference's parts machinery, FastGA.c:5057-5204, laid over devices); the output does not depend on it
 * and honours its process-level contract: -v statistics on stderr and -L:<log> the same lines appended to a log file,
 * both with the reference's "Resources for phase" / "Total Resources" lines (gene_core.c:514-593); -T threads (also the
 * layout of an index that has to be built: the -T the reference hands to GIXmake, FastGA.c:4757-4760); -P / $TMPDIR must
 * name a usable directory (FastGA.c:4456-4458, 4561) although nothing is spilled to it -- seeds and sort panels stay in
 * HBM; a GDB created from a FASTA source is removed again unless -k (Clean_Exit, FastGA.c:152-196).
 * What differs: the sub-process glue is in-process -- a missing GDB / GIX is built with this library's own producers
 * (fga_fasta_to_gdb; the index on the device, or as files with -k) instead of system("FAtoGDB"/"GIXmake"), PAF / PSL are
 * written natively instead of through ALNtoPAF / ALNtoPSL.  "#[<mask>[.1ano]]" arguments name the masks of the genome
 * before them ("#" alone: the GDB's own lower-case mask); the union of a genome's masks becomes its soft mask
 * (fga_gdb_apply_masks: Read_ANO + ANO_Union), its index is built with it and soft masking is on -- what the reference's
 * usage text promises.  (The reference itself hands the masks to `GIXmake ... #<mask>` through system(), where the shell
 * reads " #<mask>" as a comment, and files a mask behind the FIRST genome under the second, FastGA.c:4568-4573: its own
 * FastGA never applies them.  The pin for this build is therefore the two-step form, `GIXmake <genome> #<mask>` followed by
 * `FastGA -M`: tests/test_mask_files.py, tests/test_mask_files_gpu.py.)  Records that tie on (aread, abpos) come in the
 * order `FastGA -T<n>` writes them (fga_order.c; FGA_TIE_ORDER=own: an order that does not depend on -T).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include <sys/time.h>
#include <sys/resource.h>

#include "fastga_amd.h"

static FILE *Log = NULL;
static int   Verbose = 0;

/* a statistics line: stderr with -v, the log with -L */
static void say(const char *fmt, ...)
{ va_list ap;
  if (Verbose)
    { va_start(ap,fmt); vfprintf(stderr,fmt,ap); va_end(ap); fflush(stderr); }
  if (Log != NULL)
    { va_start(ap,fmt); vfprintf(Log,fmt,ap); va_end(ap); fflush(Log); }
}

/* ---- the reference's resource lines: "<user>u  <system>s  <wall>w  <cpu %>" since the start / the last phase ---- */
typedef struct { struct rusage ru; struct timespec wall; } stamp;
static stamp Start, Phase;

static void stamp_now(stamp *s)
{ getrusage(RUSAGE_SELF,&s->ru);
  clock_gettime(CLOCK_MONOTONIC,&s->wall);
}

static void put_span(FILE *f, double secs, char unit)
{ const long ms = (long) (secs*1000. + .5);
  if (ms >= 60000)
    fprintf(f,"  %ld:%02ld.%03ld%c",ms/60000,(ms/1000)%60,ms%1000,unit);
  else
    fprintf(f,"  %ld.%03ld%c",ms/1000,ms%1000,unit);
}

static void resources_to(FILE *f, const stamp *from, const stamp *now, int total)
{ const double u = (now->ru.ru_utime.tv_sec - from->ru.ru_utime.tv_sec) + 1e-6*(now->ru.ru_utime.tv_usec - from->ru.ru_utime.tv_usec);
  const double s = (now->ru.ru_stime.tv_sec - from->ru.ru_stime.tv_sec) + 1e-6*(now->ru.ru_stime.tv_usec - from->ru.ru_stime.tv_usec);
  const double w = (now->wall.tv_sec - from->wall.tv_sec) + 1e-9*(now->wall.tv_nsec - from->wall.tv_nsec);
  fprintf(f,total ? "\n  Total Resources:" : "\n  Resources for phase:");
  put_span(f,u,'u'); put_span(f,s,'s'); put_span(f,w,'w');
  fprintf(f,"  %.1f%%",w > 0. ? 100.*(u+s)/w : 0.);
  if (total)
    fprintf(f,"  %ldMB",(long) (now->ru.ru_maxrss/1000000));
  fprintf(f,"\n");
  fflush(f);
}

static void resources(int total)
{ stamp now;
  stamp_now(&now);
  if (Verbose) resources_to(stderr,total ? &Start : &Phase,&now,total);
  if (Log != NULL) resources_to(Log,total ? &Start : &Phase,&now,total);
  if (!total) Phase = now;
}

/* ---- sources ---- */
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

typedef struct
  { char *root;
    int   made_gdb;        /* this run created <root>.gdb + .<root>.bps: they go again unless -k */
    int   made_gix;        /* this run created <root>.gix + .<root>.ktab.N (only with -k): they go again when the run fails */
  } source;

static source Src[2];
static int    Keep = 0;

/* what Clean_Exit does (FastGA.c:152-196): drop the GDB this run created, unless it succeeded with -k */
static void clean_exit(int status)
{ int i;
  if (!(status == 0 && Keep))
    for (i = 0; i < 2; i++)
      if (Src[i].root != NULL && Src[i].made_gdb)
        { char *p, *slash = strrchr(Src[i].root,'/');
          if (asprintf(&p,"%s.gdb",Src[i].root) >= 0) { unlink(p); free(p); }
          if (slash != NULL)
            { if (asprintf(&p,"%.*s/.%s.bps",(int) (slash-Src[i].root),Src[i].root,slash+1) >= 0) { unlink(p); free(p); } }
          else if (asprintf(&p,".%s.bps",Src[i].root) >= 0) { unlink(p); free(p); }
        }
  if (status != 0)                                   /* a created index goes as well (GIXrm -f in the reference, FastGA.c:152-196) */
    for (i = 0; i < 2; i++)
      if (Src[i].root != NULL && Src[i].made_gix)
        { char *p, *slash = strrchr(Src[i].root,'/');
          int k;
          if (asprintf(&p,"%s.gix",Src[i].root) >= 0) { unlink(p); free(p); }
          for (k = 1; k <= 64; k++)
            { int rc;
              if (slash != NULL)
                rc = asprintf(&p,"%.*s/.%s.ktab.%d",(int) (slash-Src[i].root),Src[i].root,slash+1,k);
              else
                rc = asprintf(&p,".%s.ktab.%d",Src[i].root,k);
              if (rc >= 0) { if (unlink(p) != 0) k = 65; free(p); }
            }
        }
  if (Log != NULL) fclose(Log);
  exit(status);
}

static int prepare(const char *src, source *S, int nthreads, int want_gix_files, const char *const *masks, int nmasks)
{ int isfa;
  char *r = root_of(src,&isfa);
  S->root = r;
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
        { fprintf(stderr,"FastGA: Cannot find a GDB or FASTA for %s\n",src);
          return 1;
        }
      say("\n  Creating genome data base (GDB) %s.gdb\n",r);
      S->made_gdb = 1;
      if (fga_fasta_to_gdb(fa,r,0))
        { fprintf(stderr,"FastGA: %s\n",fga_last_error());
          return 1;
        }
      free(fa);
    }
  /* index files only with -k (otherwise the index is built on the device, in HBM only); a genome with masks named gets
     them made anew with the union of the masks, as the reference's GIXmake call would (FastGA.c:4739-4776) */
  if ((!exists("%s.gix",r) || nmasks > 0) && want_gix_files)
    { fga_gdb *g;
      say("\n  Creating genome index (GIX) %s.gix\n",r);
      S->made_gix = 1;
      if (fga_gdb_open(r,&g) || (nmasks > 0 && fga_gdb_apply_masks(g,masks,nmasks)) ||
          fga_gix_build_masked(g,r,nthreads,fga_gdb_nmask(g) > 0))
        { fprintf(stderr,"FastGA: %s\n",fga_last_error());
          return 1;
        }
      fga_gdb_close(g);
    }
  return 0;
}

static int arg_int(const char *arg, const char *what, int *val)
{ char *e;
  long v = strtol(arg+2,&e,10);
  if (e == arg+2 || *e != '\0')
    { fprintf(stderr,"FastGA: -%c '%s' argument is not an integer\n",arg[1],arg+2);
      return 1;
    }
  if (v < 0)
    { fprintf(stderr,"FastGA: %s must be non-negative (%ld)\n",what,v);
      return 1;
    }
  *val = (int) v;
  return 0;
}

int main(int argc, char *argv[])
{ fga_run_params P;
  fga_run_stats S;
  char *src[2] = { NULL, NULL }, *out = NULL, *outpath = NULL;
  const char *tmpdir, *logpath = NULL;
  int nsrc = 0, paf = 0, i;
  int ngpu = 0, ndev = 0, devices[64];
  const char *mask1[64], *mask2[64];
  int nmask1 = 0, nmask2 = 0;
  int cmin = 85, cbreak = 1000;
  double ident = .7;
  char cmd[4096];
  size_t cl = 0;

  memset(&P,0,sizeof(P));
  P.freq = 10; P.align_min = 100; P.nthreads = 8;
  cmd[0] = '\0';
  for (i = 0; i < argc && cl < sizeof(cmd)-2; i++)
    cl += snprintf(cmd+cl,sizeof(cmd)-cl,"%s%s",i ? " " : "",argv[i]);
  tmpdir = getenv("TMPDIR");
  if (tmpdir == NULL) tmpdir = ".";

  for (i = 1; i < argc; i++)
    if (argv[i][0] == '-')
      switch (argv[i][1])
      { case '1':
          if (argv[i][2] != ':' || argv[i][3] == '\0')
            { fprintf(stderr,"FastGA: option -1 must be followed by :<filename>\n"); return 1; }
          out = argv[i]+3;
          break;
        case 'f': if (arg_int(argv[i],"maximum seed frequency",&P.freq)) return 1; break;
        case 'c': if (arg_int(argv[i],"minimum seed cover",&cmin)) return 1; break;
        case 's': if (arg_int(argv[i],"seed chain break threshold",&cbreak)) return 1; break;
        case 'l': if (arg_int(argv[i],"minimum alignment length",&P.align_min)) return 1; break;
        case 'T': if (arg_int(argv[i],"number of threads to use",&P.nthreads)) return 1; break;
        case 'G':                         /* not in the reference: the comparison is cut over GPUs 0 .. n-1 of the node */
          if (arg_int(argv[i],"number of GPUs to use",&ngpu)) return 1;
          if (ngpu < 1 || ngpu > 64)
            { fprintf(stderr,"FastGA: -G number of GPUs must be in [1,64]\n"); return 1; }
          break;
        case 'i':
          { char *e;
            ident = strtod(argv[i]+2,&e);
            if (e == argv[i]+2 || *e != '\0')
              { fprintf(stderr,"FastGA: -i '%s' argument is not a real number\n",argv[i]+2); return 1; }
          }
          break;
        case 'P': tmpdir = argv[i]+2; break;
        case 'L':
          if (argv[i][2] != ':' || argv[i][3] == '\0')
            { fprintf(stderr,"FastGA: option -L must be followed by :<filename>\n"); return 1; }
          logpath = argv[i]+3;
          break;
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
              { case 'v': Verbose = 1; break;
                case 'k': Keep = 1; break;
                case 'M': P.soft_mask = 1; break;
                case 'S': P.symmetric = 1; break;
                default:
                  fprintf(stderr,"FastGA: -%c is an illegal option\n",*f);
                  return 1;
              }
          }
      }
    else if (argv[i][0] == '#')
      { /* a mask of the preceding genome (FastGA.c:4568-4573): "#" alone = its implicit mask (the lower-case intervals the
           GDB carries, GIXmake.c:1829-1832), "#<mask>[.1ano]" = a ONEcode annotation file; the union of a genome's masks is
           its soft mask, its index is built anew with it, and soft masking is on (FastGA.c:4580) */
        if ((nsrc >= 2 ? nmask2 : nmask1) >= 64)
          { fprintf(stderr,"FastGA: more than 64 masks named for one genome\n"); return 1; }
        if (nsrc >= 2) mask2[nmask2++] = argv[i]+1;
        else           mask1[nmask1++] = argv[i]+1;
        P.soft_mask = 1;
      }
    else if (nsrc < 2)
      src[nsrc++] = argv[i];
    else
      nsrc = 3;
  if (nsrc == 0 || nsrc > 2)
    { fprintf(stderr,"\nUsage: FastGA [-vkMS] [-L:<log:path>] [-T<int(8)>] [-P<dir($TMPDIR)>] [<format(-paf)>]\n"
                     "              [-f<int(10)>] [-c<int(85)>] [-s<int(1000)>] [-l<int(100)>] [-i<float(.7)>]\n"
                     "              <source1:path>[<precursor>] [<source2:path>[<precursor>]]\n\n"
                     "         <format> = -paf[mxsS]* | -psl | -1:<align:path>[.1aln]\n\n"
                     "         -G<int>: cut the comparison over that many GPUs of the node (or FGA_DEVICES=<id>,<id>,...)\n\n");
      return 1;
    }
  if ((P.paf_flags & FGA_PAF_CIGAR_M) && (P.paf_flags & FGA_PAF_CIGAR_X))
    { fprintf(stderr,"FastGA: Only one of -paf[m] or -paf[x] can be set\n"); return 1; }
  if ((P.paf_flags & FGA_PAF_CS_SHORT) && (P.paf_flags & FGA_PAF_CS_LONG))
    { fprintf(stderr,"FastGA: Only one of -paf[s] or -paf[S] can be set\n"); return 1; }
  if (P.freq < 1)                                 /* any positive cutoff, like the reference (FastGA.c:4497-4499); beyond 1982 the
                                                     merge runs on its slow, window-free kernel (fga_merge.hip)          */
    { fprintf(stderr,"FastGA: The adaptive seed count cutoff must be positive\n"); return 1; }
  if (ident < .55 || ident >= 1.)
    { fprintf(stderr,"FastGA: '-i' minimum alignment similarity must be in [0.55,1.0)\n"); return 1; }
  if (P.nthreads < 1) P.nthreads = 1;
  /* records that tie on (aread, abpos) in the order the reference's FastGA -T<n> writes them (FGA_TIE_ORDER=own: by
     (bread, strand, survival), which does not depend on -T) */
  P.reference_threads = P.nthreads;
  if (getenv("FGA_TIE_ORDER") != NULL && strcmp(getenv("FGA_TIE_ORDER"),"own") == 0)
    P.reference_threads = 0;
  if (access(tmpdir,W_OK|X_OK) != 0)
    { fprintf(stderr,"FastGA: Cannot create temporary files in directory %s\n",tmpdir); return 1; }
  if (logpath != NULL && (Log = fopen(logpath,"a")) == NULL)
    { fprintf(stderr,"FastGA: Cannot open logfile %s for output\n",logpath); return 1; }

  if (out == NULL)                       /* like the reference, PAF on stdout is the default output */
    paf = 1;
  if (paf)
    P.paf_path = "-";
  P.chain_min = 2*cmin; P.chain_break = 2*cbreak;
  P.align_rate = 1.-ident;
  P.command_line = cmd;
  if (out != NULL)
    { size_t n = strlen(out);
      if (n > 5 && strcmp(out+n-5,".1aln") == 0)
        outpath = strdup(out);
      else if (asprintf(&outpath,"%s.1aln",out) < 0)
        clean_exit(1);
      P.out_path = outpath;
    }
  if (nsrc == 1 && nmask2 > 0)                    /* (cannot happen: masks after the only genome are its own) */
    nmask2 = 0;
  for (i = 0; i < nsrc; i++)
    if (prepare(src[i],Src+i,P.nthreads,Keep,i == 0 ? mask1 : mask2,i == 0 ? nmask1 : nmask2))
      clean_exit(1);
  P.masks1 = mask1; P.nmasks1 = nmask1;
  P.masks2 = mask2; P.nmasks2 = nmask2;
  if (nsrc == 2 && strcmp(Src[0].root,Src[1].root) == 0)
    nsrc = 1;

  stamp_now(&Start);
  Phase = Start;
  if (Log != NULL) fprintf(Log,"\n%s\n",cmd);
  /* the devices of the run: -G<n> = GPUs 0 .. n-1, else FGA_DEVICES=<id>,<id>,... (ids may repeat: ranks sharing a GPU),
     else GPU 0.  More than one: fga_run_multi, one host thread + stream per device, seeds exchanged by A-contig part */
  if (ngpu > 0)
    for (ndev = 0; ndev < ngpu; ndev++)
      devices[ndev] = ndev;
  else if (getenv("FGA_DEVICES") != NULL && getenv("FGA_DEVICES")[0] != '\0')
    { const char *e = getenv("FGA_DEVICES");
      while (*e != '\0')
        { char *end;
          long v = strtol(e,&end,10);
          if (end == e || v < 0 || v > 1023 || (*end != ',' && *end != '\0') || ndev >= 64)
            { fprintf(stderr,"FastGA: FGA_DEVICES must be a comma-separated list of at most 64 GPU numbers\n");
              clean_exit(1);
            }
          devices[ndev++] = (int) v;
          e = (*end == ',') ? end+1 : end;
        }
    }
  if (ndev == 0)
    devices[ndev++] = P.device;
  P.device = devices[0];
  if (ndev == 1)
    say("\n  Using GPU %d and %d host threads\n",P.device,P.nthreads);
  else
    { say("\n  Using %d GPUs (",ndev);
      for (i = 0; i < ndev; i++) say("%s%d",i ? "," : "",devices[i]);
      say(") and %d host threads\n",P.nthreads);
    }
  if (fga_run_multi(Src[0].root,nsrc == 2 ? Src[1].root : NULL,&P,ndev,devices,&S))
    { fprintf(stderr,"FastGA: %s\n",fga_last_error());
      clean_exit(1);
    }
  say("\n  Total seeds = %lld, ave. len = %.1f, seeds per G1 position = %.1f\n",(long long) S.nseeds,
      S.nseeds ? (1.*S.seed_len_sum)/S.nseeds : 0.,S.bases1 > 0 ? (1.*S.nseeds)/S.bases1 : 0.);
  say("  Phase 1 on the device: merge %.3fs (kernel %.3f ms); index + genomes to HBM %.3fs + %.3fs\n",
      S.merge_s,S.merge_kernel_ms,S.load_s,S.upload_s);
  say("\n  Total hits over %dbp = %lld, %lld aln's, %lld non-redundant aln's of ave len %lld\n",
      cmin,(long long) S.nhits,(long long) S.nalns,(long long) S.nlive,(long long) (S.nlive ? S.cover/S.nlive : 0));
  say("  Phase 2 on the device (%d part%s): sort %.3fs (kernel %.3f ms) chain %.3fs extend %.3fs (kernel %.3f ms, %lld calls,"
      " %lld waves) filter %.3fs write %.3fs\n",S.nparts,S.nparts == 1 ? "" : "s",
      S.sort_s,S.sort_kernel_ms,S.chain_s,S.extend_s,S.extend_kernel_ms,
      (long long) S.ncalls,(long long) S.nwaves,S.filter_s,S.write_s);
  if (paf)
    say("  %s: edit scripts %.3fs (kernels %.3f ms), regroup + format %.3fs\n",(P.paf_flags & FGA_OUT_PSL) ? "PSL" : "PAF",
        S.trace_s,S.trace_kernel_ms,S.paf_s);
  resources(0);
  resources(1);
  free(outpath);
  clean_exit(0);
  return 0;
}
