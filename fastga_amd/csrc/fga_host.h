/* fga_host.h -- internal host-side types shared by the C sources of libfastga_amd.
 *
 * On-disk formats follow the reference exactly:
 *   GDB  : <root>.gdb (ASCII ONEcode) or <root>.1gdb + .<root>.bps       (reference GDB.c:1181-1410)
 *   GIX  : <root>.gix + .<root>.ktab.<p>                                  (reference GIXmake.c:1490-1580)
 */
#ifndef FGA_HOST_H
#define FGA_HOST_H

#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGA_KMER      40          /* only k = 40 indices are supported (SURVEY.md hard part 8) */
#define FGA_NPREFIX   0x1000000   /* 2^24 12-mer prefixes                                       */

typedef struct
  { int64_t clen;      /* contig length in bases                         */
    int64_t sbeg;      /* start of contig inside its scaffold            */
    int64_t boff;      /* byte offset of the contig in the .bps image    */
    int     scaf;      /* scaffold index                                  */
  } fga_contig;

typedef struct
  { int64_t slen;
    int     fctg, ectg;
    int64_t hoff;      /* offset of the header string in headers[]       */
  } fga_scaffold;

struct fga_gdb
  { int           nscaff;
    fga_scaffold *scaffolds;
    int           ncontig;
    fga_contig   *contigs;
    int64_t       maxctg;
    int64_t       seqtot;
    int64_t       hdrtot;
    char         *headers;
    float         freq[4];
    char         *path;      /* path of the skeleton file as opened      */
    char         *srcpath;   /* reference line of the skeleton           */
    uint8_t      *bps;       /* whole 2-bit image (base i of a contig in bits 2*(i&3) of byte i>>2) */
    int64_t       bpslen;
    /* soft mask (lower-case runs of the FASTA), contig coordinates, from .<root>.msk when present:
       intervals [mbeg[i],mend[i]) for i in [moff[c],moff[c+1])                                        */
    int64_t       nmask;
    int64_t      *moff;      /* [ncontig+1] */
    int64_t      *mbeg, *mend;
  };

struct fga_gix
  { int       kmer;        /* 40                                                        */
    int       nparts;      /* number of .ktab parts                                     */
    int       postbytes;   /* bytes of the in-contig position                           */
    int       contbytes;   /* bytes of the (length-sorted) contig index + sign bit      */
    int       ebytes;      /* 9 + postbytes + contbytes                                 */
    int64_t   maxpre;      /* largest 12-mer panel                                      */
    int       freq;
    int       nctg;
    int      *perm;        /* length-sorted -> original contig index                    */
    int64_t  *index;       /* [2^24] inclusive cumulative entry count per 24-bit prefix */
    int64_t   nents;
    uint8_t  *table;       /* nents * ebytes raw on-disk entries, parts concatenated    */
    int64_t  *partbeg;     /* [nparts+1] entry offset of each part                      */
    int       legacy;      /* read from the pre-v1.3 layout: k-mers above `freq` positions are not in it */
  };

typedef struct fga_gdb fga_gdb;
typedef struct fga_gix fga_gix;

/* error reporting: thread-local message buffer; functions return 0 on success */
void        fga_set_error(const char *fmt, ...);
const char *fga_last_error(void);

/* GDB (fga_gdb.c) */
int      fga_fasta_to_gdb(const char *fasta, const char *target, int ncut);
int      fga_gdb_open(const char *path, fga_gdb **out);
void     fga_gdb_close(fga_gdb *G);
uint8_t *fga_gdb_get_contig(const fga_gdb *G, int c, uint8_t *buf);
int      fga_gdb_write_skeleton(const fga_gdb *G, const char *path, const char *prog, const char *command);
int      fga_gdb_apply_masks(fga_gdb *G, const char *const *paths, int npaths);   /* "#<mask>" arguments: the union becomes G's soft mask */

/* binary ONEcode pieces shared by the GDB and the .1aln readers (fga_one.c) */
typedef struct
  { int      have;
    int      esc, esclen;
    uint8_t  len[256];
    uint8_t *look;           /* 65536 entries: symbol of every 16-bit prefix */
  } fga_one_codec;
int     fga_one_int(const uint8_t *u, const uint8_t *end, int64_t *val);
int     fga_one_codec_parse(fga_one_codec *c, const uint8_t *in, int64_t n);
int64_t fga_one_codec_decode(const fga_one_codec *c, const uint8_t *in, int64_t nbits, uint8_t *out, int64_t cap);
int     fga_one_footer_codecs(const uint8_t *buf, size_t size, fga_one_codec *codec /* [128] */);

/* GIX (fga_gix.c) */
int      fga_gix_open(const char *path, fga_gix **out);
void     fga_gix_close(fga_gix *X);
int      fga_gix_build(const fga_gdb *G, const char *target, int nthreads);
int      fga_gix_layout(const fga_gdb *G, int nthreads, int *nctg, int **perm, int **invp,
                        int *postbytes, int *contbytes, int *nparts);
void     fga_gix_ksplit(const int64_t *sbuck, int nparts, int *ksplit);
const uint8_t *fga_gix_tmap(void);
int      fga_gix_write_files(const fga_gix *X, const char *target);

/* device context internals the host pipeline uses (fga_device.hip) */
struct fga_dev;
void   fga_dev_trim(struct fga_dev *dev);          /* unused regions of the device pool back to the driver */
void  *fga_dev_stage_acquire(struct fga_dev *dev, size_t bytes);   /* the per-part staging buffer of a multi-pass run (a workspace slot) */
void   fga_dev_stage_release(struct fga_dev *dev, void *ptr);
size_t fga_dev_available(struct fga_dev *dev);     /* free device memory + the pool's free pieces    */
int    fga_dev_current_device(void);               /* the calling thread's current HIP device (-1: cannot be told) */
void   fga_dev_restore_device(int device);         /* ... and back to it (an entry point that visits several devices) */

void   fga_aln_writer_threads(int n);              /* threads of the .1aln record formatters, for the calling thread (default 8) */
void   fga_note(const char *what, double since);   /* FGA_TIMING=1: elapsed wall time since `since` on stderr */

/* the library's large host arrays are kept between comparisons (fga_hbuf.c): blocks of 4 MB and more come from a cache of
   their own and go back to it; everything else is plain malloc / free.  The C sources reach it through these macros */
void  *fga_big_malloc(size_t n);
void  *fga_big_calloc(size_t n, size_t m);
void  *fga_big_realloc(void *p, size_t n);
void   fga_big_free(void *p);
void   fga_host_cache_trim(void);                  /* the parked blocks back to the system */
#if !defined(FGA_NO_MALLOC_MACROS) && !defined(__cplusplus)
#define malloc(n)     fga_big_malloc(n)
#define calloc(n,m)   fga_big_calloc(n,m)
#define realloc(p,n)  fga_big_realloc(p,n)
#define free(p)       fga_big_free(p)
#endif

/* small helpers */
char *fga_path_dir(const char *path);                       /* malloc'd directory part ("." if none) */
char *fga_path_root(const char *path, const char *suffix);  /* malloc'd basename without suffix      */
double fga_wall(void);

/* a team of threads for the O(n) passes of one host stage (fga_par.c) */
typedef struct fga_team fga_team;
typedef void (*fga_slice_fn)(void *arg, int slice, int64_t begin, int64_t end);
fga_team *fga_team_open(int nthreads);
int       fga_team_size(const fga_team *T);
void      fga_team_run(fga_team *T, int64_t n, fga_slice_fn fn, void *arg);    /* [0,n) in one contiguous slice per thread */
int       fga_team_sort_pairs(fga_team *T, uint64_t *key, int64_t *val, int64_t n, int bits);   /* stable LSD radix sort */
void      fga_team_close(fga_team *T);

#ifdef __cplusplus
}
#endif
#endif
