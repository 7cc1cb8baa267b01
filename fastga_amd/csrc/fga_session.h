/* fga_session.h -- the session object of fga_pipeline.c, shared with fga_multi.c (one comparison over several GPUs from
 * one process).  Internal: the C-ABI only ever hands out the opaque pointer. */
#ifndef FGA_SESSION_H
#define FGA_SESSION_H

#include "fga_host.h"
#include "fastga_amd.h"

struct fga_session
  { fga_gdb *g1, *g2;
    fga_gix *x1, *x2;
    fga_dev *dev;
    fga_dgix *d1, *d2;
    fga_dgenome *dg1, *dg2;
    int self;
    int devbuilt;              /* an index was built on the device (no soft-mask bytes in it) */
    double load_s, upload_s;
    int nranks, rank;          /* > 1: the session holds the rank's 12-mer prefix range of both tables only */
    int64_t *cuts;             /* [nranks+1] the prefix ranges of all ranks */
    int64_t *scount;           /* [2*nctg] seeds per (strand, A contig) of the merges so far (reference order only) */
    int borrowed_gdb;          /* g1 / g2 belong to the caller (fga_run_multi opens them once for all its ranks) */
    int borrowed_gix;          /* x1 / x2 (index files read once) likewise; an index a rank builds on the device is its own */
  };

typedef struct { const char *const *m1; int n1; const char *const *m2; int n2; } fga_mask_args;

/* host-side inputs opened once and lent to every rank's session (NULL members: the session opens its own) */
typedef struct { fga_gdb *g1, *g2; fga_gix *x1, *x2; } fga_shared_inputs;

int fga_session_open_impl(const char *root1, const char *root2, int device, int nthreads, int rank, int nranks, int flags,
                          const fga_mask_args *masks, const fga_shared_inputs *shared, fga_session **out);
int fga_gix_files_exist(const char *root);

/* the .1aln as a stream of A-contig stretches (fga_session_run's parts, fga_multi_run's ranks): whether this run's output
   can be one, the stream on the session's genomes, and a filtered set put into the reference's tie order */
int fga_run_can_stream(const fga_run_params *P);
int fga_run_stream_forced(void);

#endif
