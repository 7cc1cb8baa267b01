/* fastga_amd.h -- C-ABI of libfastga_amd.so: the MI355X-native seed-and-extend hot path of FastGA.
 *
 * FastGA itself has no plugin / FFI interface: its seams are the `FastGA` process, the
 * cross-file C functions `rmsd_sort` / `Local_Alignment`, and the file-static pipeline stages of
 * FastGA.c.  Every entry point below names the reference interface it replaces (file:line under
 * /root/reference).  All functions are re-entrant, take plain pointers and sizes, never call exit(), and
 * return 0 on success / non-zero on failure with a message available from fga_last_error().
 */
#ifndef FASTGA_AMD_H
#define FASTGA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fga_gdb fga_gdb;    /* genome database: skeleton + 2-bit bases (host)        */
typedef struct fga_gix fga_gix;    /* genome index: 12-mer prefix index + k-mer table (host) */

const char *fga_last_error(void);

/* ---- GDB: replaces FAtoGDB / Create_GDB (GDB.c:442-1170) and Read_GDB / Get_Contig (GDB.c:1181, 1739) -- */
int      fga_fasta_to_gdb(const char *fasta, const char *target, int ncut);
int      fga_gdb_open(const char *path, fga_gdb **out);
void     fga_gdb_close(fga_gdb *gdb);
int      fga_gdb_ncontig(const fga_gdb *gdb);
int      fga_gdb_nscaff(const fga_gdb *gdb);
int64_t  fga_gdb_seqtot(const fga_gdb *gdb);
int64_t  fga_gdb_maxctg(const fga_gdb *gdb);
int64_t  fga_gdb_contig_len(const fga_gdb *gdb, int contig);
void     fga_gdb_freq(const fga_gdb *gdb, float *freq4);
uint8_t *fga_gdb_get_contig(const fga_gdb *gdb, int contig, uint8_t *buf /* clen+2 bytes */);

/* ---- GIX: replaces GIXmake (GIXmake.c:1635-2058) and Open_Kmer_Stream + Open_Post_List header parse
 *      (libfastk.c:785-907, FastGA.c:283-333) ------------------------------------------------------------- */
int            fga_gix_build(const fga_gdb *gdb, const char *target, int nthreads);
int            fga_gix_build_masked(const fga_gdb *gdb, const char *target, int nthreads, int use_mask); /* GIXmake ... # */
int64_t        fga_gdb_nmask(const fga_gdb *gdb);   /* soft-mask (lower-case) intervals carried by the GDB */
int            fga_gix_open(const char *path, fga_gix **out);
void           fga_gix_close(fga_gix *gix);
int64_t        fga_gix_nents(const fga_gix *gix);
int            fga_gix_ebytes(const fga_gix *gix);
int            fga_gix_postbytes(const fga_gix *gix);
int            fga_gix_contbytes(const fga_gix *gix);
int            fga_gix_nctg(const fga_gix *gix);
int            fga_gix_nparts(const fga_gix *gix);
int64_t        fga_gix_part_begin(const fga_gix *gix, int part);   /* first entry of table part (0..nparts) */
int64_t        fga_gix_maxpre(const fga_gix *gix);
const int     *fga_gix_perm(const fga_gix *gix);
int            fga_gix_legacy_cutoff(const fga_gix *gix);          /* 0: today's layout; else the index came from the
                                                                      pre-v1.3 layout (.post.N files, FastGA.c:206-570) and holds
                                                                      no k-mer with more positions than this: -f must not exceed it */
const int64_t *fga_gix_index(const fga_gix *gix);
const uint8_t *fga_gix_table(const fga_gix *gix);

/* ==== device side (MI355X / gfx950) =================================================================== */

typedef struct fga_dev    fga_dev;     /* one GPU: device id, HIP stream, timing events            */
typedef struct fga_dgix   fga_dgix;    /* device-resident genome index (one array per field + prefix index) */
typedef struct fga_dseeds fga_dseeds;  /* device-resident adaptive seeds                            */

/* One adaptive seed.  Carries exactly the fields of the reference's seed temp record
 * {u8 plen; A post|contig; B post|contig|sign} (FastGA.c:961-966) in a fixed 16-byte layout. */
typedef struct
  { uint32_t apos;     /* A in-contig position as stored in the index payload                          */
    uint32_t bpos;     /* B in-contig position                                                         */
    uint32_t actg;     /* (A contig, length-sorted index) << 8 | plen                                  */
    uint32_t bctg;     /* B contig | (B entry's own sign bit) << 30 | (1 << 31 if complement stream)   */
  } fga_seed;

typedef struct
  { int     freq;          /* -f : adaptamer frequency cutoff (FastGA.c:4451); any positive value (beyond 1982: the slow, window-free kernel) */
    int     soft_mask;     /* -M or #mask arguments: mlen = plen (FastGA.c:824-825)                     */
    int     flip;          /* second pass of -S: table 1 is genome 2 (FastGA.c:2410-2470)               */
    int64_t prefix_begin;  /* 12-mer prefix range [begin,end) handled by this call (multi-GPU sharding; */
    int64_t prefix_end;    /*   the reference splits threads the same way, FastGA.c:2291-2321); 0,0=all */
  } fga_merge_params;

enum { FGA_STAGE_MERGE_PARTITION = 0, FGA_STAGE_MERGE = 1, FGA_STAGE_SORT = 2, FGA_STAGE_CHAIN = 3,
       FGA_STAGE_EXTEND = 4, FGA_STAGE_GIX = 5, FGA_STAGE_TRACE = 6, FGA_STAGE_REGROUP = 7, FGA_NSTAGES = 8 };

int   fga_dev_open(int device, fga_dev **out);
void  fga_dev_close(fga_dev *dev);        /* not while another thread can still allocate, copy or launch under the context */
int   fga_dev_sync(fga_dev *dev);
float fga_dev_stage_ms(const fga_dev *dev, int stage);   /* HIP-event time of the stage's last launch */

/* plain device memory for callers that stage records themselves (exchange buffers of a multi-GPU run when the caller does
   not bring its own allocator); a pointer from any allocator of the same HIP runtime is equally acceptable wherever this
   header says "device buffer" */
int   fga_dev_malloc(fga_dev *dev, size_t bytes, void **out);
void  fga_dev_free(fga_dev *dev, void *ptr);
int   fga_dev_download(fga_dev *dev, void *host_dst, const void *device_src, size_t bytes);
int   fga_dev_upload(fga_dev *dev, void *device_dst, const void *host_src, size_t bytes);
double  fga_dev_driver_seconds(void);       /* seconds this process has waited in hipMalloc / hipFree for the pool's regions: 0.3 ms
                                               a call, unless the driver first has to clear memory another process released (seconds) */
int64_t fga_dev_peak_bytes(fga_dev *dev);   /* peak device memory in use by this process so far (stage-boundary samples) */
/* Device memory of a MiB and more is cut from regions the library keeps for reuse (allocation and release of tens of GB
   through the driver cost seconds).  fga_dev_trim gives the regions nobody uses back to the driver -- call it before another
   library of the process (an RCCL / torch allocator) needs the room; fga_dev_available = free device memory + what the
   library's free pieces hold. */
void    fga_dev_trim(fga_dev *dev);
size_t  fga_dev_available(fga_dev *dev);
void    fga_dev_set_host_threads(fga_dev *dev, int nthreads);   /* threads for the host tails of the device stages (unit
                                                                   order, hit re-lay, work order of the extension); default 1 */

int   fga_dgix_upload(fga_dev *dev, const fga_gix *gix, fga_dgix **out);
void  fga_dgix_free(fga_dgix *dgix);
int64_t fga_dgix_nents(const fga_dgix *dgix);      /* entries resident on the device (a slice: of its prefix range) */
/* The index built on the device straight into HBM: replaces the GIXmake run for the seed merge's input (GIXmake.c
 * sample / distribution / sort / merge threads).  `gdb` must hold its bases (fga_gdb_open); nthreads plays GIXmake's
 * -T for the layout (contig padding, table parts).  *dgix is what fga_dgix_upload of the fga_gix_build files would
 * give, byte for byte; *gix is the matching host descriptor (perm, widths; with FGA_GIX_HOST_COPY also index + table,
 * for writing the files or for tests). */
#define FGA_GIX_HOST_COPY 1     /* keep index + table on the host as well (fga_gix_write_files, tests)          */
#define FGA_GIX_SOFT_MASK 2     /* fill the soft-mask byte from the GDB's lower-case intervals (GIXmake's '#')  */
int   fga_dgix_build(fga_dev *dev, const fga_gdb *gdb, int nthreads, int flags, fga_dgix **dgix, fga_gix **gix);
/* <root>.gix + .<root>.ktab.N from an index that holds a host copy of its table (fga_dgix_build with want_host_copy,
 * or one loaded with fga_gix_open): the files fga_gix_build / GIXmake write */
int   fga_gix_write_files(const fga_gix *gix, const char *target);
/* One rank's slice of a table: only the entries whose 12-mer prefix lies in [pbeg,pend) (the reference's merge
 * threads each read one such range of both tables, FastGA.c:2291-2321) -- read from index files, or built on the device
 * from the GDB with 1/N of the sort.  fga_dgix_prefix_counts gives the per-prefix entry counts of the table a build would
 * make (one syncmer scan, nothing built): what the ranges are cut from (fga_session_open_sliced does all of it). */
int   fga_dgix_upload_range(fga_dev *dev, const fga_gix *gix, int64_t pbeg, int64_t pend, fga_dgix **out);
int   fga_dgix_build_range(fga_dev *dev, const fga_gdb *gdb, int nthreads, int flags, int64_t pbeg, int64_t pend,
                           fga_dgix **dgix, fga_gix **gix);
/* fga_dgix_build (pbeg = 0, pend = 2^24) / fga_dgix_build_range that leaves the genome's bases on the device, laid out for
 * fga_dgenome_adopt: they cross PCIe once per session */
int     fga_dgix_build_keep(fga_dev *dev, const fga_gdb *gdb, int nthreads, int flags, int64_t pbeg, int64_t pend,
                            fga_dgix **out, fga_gix **host_meta, void **image);
int   fga_dgix_prefix_counts(fga_dev *dev, const fga_gdb *gdb, int nthreads, uint32_t *counts /* host, 2^24 */);

/* Adaptive seed merge: replaces adaptamer_merge -> new_merge_thread (FastGA.c:2281, 610) and, with
 * t2 == NULL, self_adaptamer_merge -> new_self_merge_thread (FastGA.c:2496, 1616).  Returns 0, or 2 when
 * more than `capacity` seeds were found (count is still exact; re-run with a larger buffer).
 * capacity <= 0 picks 2 x (entries of t1 in range) + 1M.
 * t1 keeps the range cuts of its last launch (reused when the next one names the same tables -- by their view generations --,
 * prefix range and geometry): a table must not be table 1 of two merges running at the same time (two sessions never share
 * one; a caller of the stage API that does serialises those launches)                                 */
int     fga_seed_merge(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2,
                       const fga_merge_params *prm, int64_t capacity, fga_dseeds **out);
int64_t fga_seeds_count(const fga_dseeds *seeds);
int64_t fga_seeds_plen_sum(const fga_dseeds *seeds);   /* the reference's "ave. len" numerator */
int     fga_seeds_download(const fga_dseeds *seeds, fga_seed *host, int64_t max);
void    fga_seeds_free(fga_dseeds *seeds);

/* Seed -> diagonal record transform + sort: replaces reimport_thread (FastGA.c:2641-2747) and rmsd_sort
 * (RSDsort.c:292).  Result: 128-bit keys, ascending, fields packed most-significant first
 * [strand | A contig | B contig | diag>>6 | anti | diag&63 | lcp] with the bit widths of fga_keys_layout. */
typedef struct fga_dkeys fga_dkeys;
typedef struct
  { int64_t amxpos, bmxpos;    /* longest contig of genome 1 / genome 2 (FastGA.c:5020-5041)  */
    int     nctg_a, nctg_b;    /* contig counts (bounds of the contig fields)                 */
    int     anti_order_only;   /* 1: leave records of equal (strand, contigs, diag>>6, anti) in arrival order, i.e. do
                                  not sort on diag&63 and lcp (two radix passes fewer).  The chain scan's result does not
                                  depend on the order inside such a tie; 0 gives the reference's full record order.   */
  } fga_sort_params;

int     fga_seed_sort(fga_dev *dev, const fga_dseeds *seeds, const fga_sort_params *prm, fga_dkeys **out);
int64_t fga_keys_count(const fga_dkeys *keys);
void    fga_keys_layout(const fga_dkeys *keys, int *wa, int *wb, int *wd, int *wt);
int     fga_keys_download(const fga_dkeys *keys, void *host /* 16 B per key: lo64, hi64 */, int64_t max);
const void *fga_keys_download_pinned(const fga_dkeys *keys);  /* into the device context's pinned staging buffer */
void    fga_keys_free(fga_dkeys *keys);
/* rmsd_sort (RSDsort.c:292) on records that already sit in HBM: n 128-bit little-endian records in the device buffer
 * buf0, ordered ascending on bits [lowbit, lowbit+nbits) -- stable, LSD radix --, buf1 = scratch of the same size; *sorted
 * is whichever of the two holds the result.  Enqueued on the context's stream and synchronised before return. */
int     fga_dev_radix_sort_u128(fga_dev *dev, void *buf0, void *buf1, int64_t n, int lowbit, int nbits, void **sorted);

/* ---- chain detection: replaces the chain scan of align_contigs (FastGA.c:3016-3176, 3340-3403) ------------ */
typedef struct
  { int32_t dgmin, dgmax;      /* diagonal range of the chain ("tube"), contig coordinates (FastGA.c:3205-3216) */
    int64_t alow, ahgh;        /* anti-diagonal range                                                           */
    int32_t cov, pad;
  } fga_hit;                   /* 32 bytes */

typedef struct
  { int32_t actg, bctg;        /* length-sorted contig indices                    */
    int32_t comp;              /* 1: complement stream                            */
    int32_t nhits;
    int64_t first_hit;         /* index of the unit's first hit                   */
    int64_t bucket;            /* diag>>6 of the unit's first bucket              */
  } fga_unit;                  /* 32 bytes; a unit = bucket pair (d,d+1) of one contig pair and strand */

typedef struct
  { int64_t chain_break;       /* -s, doubled (FastGA.c:4547)                     */
    int64_t chain_min;         /* -c, doubled (FastGA.c:4495)                     */
    int64_t amxpos, bmxpos;
    const int64_t *alen;       /* A contig lengths by length-sorted index         */
    int64_t nalen;             /* entries in alen (the device scan copies them)   */
  } fga_chain_params;

typedef struct
  { int64_t   nhits, nunits;
    fga_hit  *hits;
    fga_unit *units;
  } fga_hits;

int  fga_chain_scan(const void *keys /* n x {lo64,hi64} */, int64_t n, int wa, int wb, int wd, int wt,
                    const fga_chain_params *prm, int nthreads, fga_hits **out);
/* the same scan on the device-resident keys: the 16 B/record stream stays in HBM, only the hits come back.
 * Units are returned in the order of fga_chain_scan (by first record), hits contiguous per unit. */
int  fga_chain_scan_device(fga_dev *dev, const fga_dkeys *keys, const fga_chain_params *prm, fga_hits **out);
int  fga_hits_create(const fga_unit *units, int64_t nunits, const fga_hit *hits, int64_t nhits, fga_hits **out);
void fga_hits_free(fga_hits *hits);
int64_t         fga_hits_count(const fga_hits *hits);
int64_t         fga_hits_nunits(const fga_hits *hits);
const fga_hit  *fga_hits_array(const fga_hits *hits);
const fga_unit *fga_hits_units(const fga_hits *hits);

/* ---- wave extension: replaces the per-hit loop of align_contigs (FastGA.c:3227-3341) around Local_Alignment
 *      (align.h:235-236), New_Align_Spec (align.h:196) and the accept test + Compress_TraceTo8 (FastGA.c:3264-3281) */
typedef struct fga_dgenome fga_dgenome;    /* device-resident 2-bit genome (+ reverse-complement image) */

typedef struct
  { int32_t  tlen, diffs, abpos, bbpos, aepos, bepos;   /* Path (align.h:89-95), trace as tlen bytes  */
    uint32_t flags;                                      /* COMP_FLAG (0x1) for the complement stream  */
    int32_t  aread, bread;                               /* original contig indices                    */
    int32_t  unit, seq;                                  /* provenance: unit index, order inside it    */
    int32_t  pad;
    int64_t  toff;                                       /* offset of the trace bytes                  */
  } fga_aln;                                             /* 56 bytes */

typedef struct
  { int     tspace;            /* 100 (FastGA.c:46)                                              */
    int     path_ave;          /* Align_Spec.ave_path (align.c:251)                              */
    const int16_t *table;      /* 32768 entries each (align.c:207-218)                           */
    const int16_t *score;
    int     self;              /* comparing a genome against itself                              */
    int     aln_min;           /* ALIGN_MIN - 50 (FastGA.c:3013)                                 */
    double  aln_rate;          /* ALIGN_RATE + .05 (FastGA.c:3014)                               */
    int64_t cell_cap;          /* trace-point cells per wavefront (0: derived from contig size)  */
    int64_t aln_cap, trace_cap;/* output capacities (0: derived from the hit count)              */
  } fga_extend_params;

typedef struct
  { int64_t  naln, ntrace, ncalls, nwaves;
    fga_aln *alns;             /* host copies */
    uint8_t *tbytes;
    /* accounting of the extension launch that produced the set (B_ext = 2 (nbases + ncells) + 2 ntrace) */
    int64_t  ncells;           /* diagonal updates: sum over wave steps of the wave width                   */
    int64_t  nbases;           /* bases compared by the snakes (per sequence; + one probe per cell update)  */
    double   busy_waves;       /* wavefronts busy on average over the launch                                */
    int64_t *ctg_waves;        /* [nctg_waves] wave steps per A contig (index order) of that launch, or NULL: the cost  */
    int      nctg_waves;       /*   fga_multi_run weighs the next partition of the contigs over its ranks with          */
  } fga_alns;

int  fga_align_spec(double ave_corr, int tspace, const float *freq4, int *path_ave, int16_t *table, int16_t *score);
int  fga_dgenome_upload(fga_dev *dev, const fga_gdb *gdb, const int *perm, int nperm, int want_revcomp,
                        fga_dgenome **out);
/* the same over bases that are on the device already: `image` is what fga_dgix_build_keep returned for the same GDB (bpslen
 * bytes between two zero pads); the genome owns it from the call on, also when the call fails */
int     fga_dgenome_adopt(fga_dev *dev, const fga_gdb *gdb, const int *perm, int nperm, int want_revcomp, void *image,
                          fga_dgenome **out);
void fga_dgenome_free(fga_dgenome *g);
int  fga_extend(fga_dev *dev, const fga_dgenome *ga, const fga_dgenome *gb, const fga_hits *hits,
                const fga_extend_params *prm, fga_alns **out);
void fga_alns_free(fga_alns *alns);

/* second pass of -S appended to an existing seed buffer (FastGA.c:2410-2470); buffer must have room */
int  fga_seed_merge_append(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2, const fga_merge_params *prm,
                           fga_dseeds *seeds);

/* ---- redundancy filter + final order (FastGA.c:3405-3694, 3800-3835) and .1aln emission (alncode.c:239-305) ---- */
int  fga_filter_alignments(const fga_alns *in, fga_alns **out);
int  fga_filter_alignments_mt(const fga_alns *in, int nthreads, fga_alns **out);   /* same result, contig pairs in parallel */
int  fga_write_1aln(const char *path, const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns,
                    int tspace, const char *db1_name, const char *db2_name, const char *command_line);
/* the same content in the binary ONEcode container the reference writes (object index in the footer, so the
 * reference's seeking readers -- ALNtoPAF, ALNshow -- accept it); fga_write_1aln is the ASCII form */
int  fga_write_1aln_binary(const char *path, const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns,
                           int tspace, const char *db1_name, const char *db2_name, const char *command_line);
/* the same file as a stream: header and skeletons at open, record sets appended in final order (a comparison that runs phase 2
 * in several parts writes a part's records while the next part's kernels run), footer -- counts and object indices -- at close
 * (keep = 0: the file is removed).  What the reference's per-thread Write_Aln_* calls into one file are (alncode.c:239-305) */
typedef struct fga_aln_stream fga_aln_stream;
int  fga_aln_stream_open(const char *path, const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, int tspace,
                         const char *db1_name, const char *db2_name, const char *command_line, fga_aln_stream **out);
int  fga_aln_stream_append(fga_aln_stream *s, const fga_alns *alns);
/* append in two steps, for writers that hold their set before its turn in the file has come: the records formatted by any
   thread (the stream is read, not changed; fga_aln_stream_preformats: not when list codes are wanted, FGA_ALN_CODEC=1 -- those
   are trained on the file's first set), the block committed in the file's order (consumed, also on failure) */
typedef struct fga_aln_block fga_aln_block;
int  fga_aln_stream_preformats(const fga_aln_stream *s);
int  fga_aln_stream_format(const fga_aln_stream *s, const fga_alns *alns, fga_aln_block **out);
int  fga_aln_stream_commit(fga_aln_stream *s, fga_aln_block *block);
void fga_aln_block_free(fga_aln_block *block);
int64_t fga_aln_stream_records(const fga_aln_stream *s);
int  fga_aln_stream_close(fga_aln_stream *s, int keep);

/* ---- edit scripts from trace points: replaces Compute_Trace_PTS (align.h:341-342, align.c:6171-6308) in the mode every
 *      reader of a .1aln uses it (ALNtoPAF.c:278, ALNshow.c:524, ALNtoPSL.c:193, ONEaln.c:1011: GREEDIEST, dlow > dhgh)
 *      for all alignments of a set at once.  The result is Path.trace / Path.tlen / Path.diffs of each alignment after
 *      the call: one int per indel, -(p) = gap in A before its p-th base, +(q) = gap in B before its q-th base (1-based
 *      contig coordinates, B complemented when COMP_FLAG is set).  genome 2 needs its reverse-complement image when
 *      the set holds complement alignments.  self != 0 reproduces the reference's aseq == bseq rule (align.c:6256-6265)
 *      for same-contig forward alignments; the reference's readers load A and B into separate buffers, i.e. self = 0. */
typedef struct
  { int64_t  naln, ntrace, npanels;
    int64_t *toff;             /* [naln+1] offset of each alignment's ints in trace */
    int32_t *tlen;             /* [naln]   Path.tlen after Compute_Trace_PTS        */
    int32_t *diffs;            /* [naln]   Path.diffs after Compute_Trace_PTS       */
    int32_t *trace;            /* [ntrace]                                          */
    int32_t *resume;           /* NULL: the scripts are Compute_Trace_PTS's.  Else [naln], scripts of
                                  fga_trace_pts_regrouped: -1 = Gap_Improver has been applied to the alignment, x >= 0 =
                                  applied to the entries before x, the box starting at x did not fit a lane's scratch
                                  (or the alignment is too long for one lane): the host regrouping continues from x  */
  } fga_traces;

int  fga_trace_pts(fga_dev *dev, const fga_dgenome *ga, const fga_dgenome *gb, const fga_alns *alns,
                   int tspace, int self, fga_traces **out);
/* Compute_Trace_PTS followed by Gap_Improver (align.h:393-399, align.c:6714-7133) on the device, as every reader of a
 * .1aln calls the two back to back (ALNtoPAF.c:278-283, ALNtoPSL.c:193-197): one lane regroups one alignment's script in
 * place, reading the sequences from the packed genome images (A: the whole contig, B: the aligned piece, ALNtoPAF.c:
 * 258-277); trace / diffs are Path.trace / Path.diffs after Gap_Improver wherever resume[i] < 0.  fga_write_paf /
 * fga_write_psl / fga_gap_improve accept the result and finish the alignments the device handed back. */
int  fga_trace_pts_regrouped(fga_dev *dev, const fga_dgenome *ga, const fga_dgenome *gb, const fga_alns *alns,
                             int tspace, int self, fga_traces **out);
void fga_traces_free(fga_traces *t);
/* fga_trace_pts_regrouped's per-alignment routine (fga_gapcore.inc, the source the device kernel is compiled from)
 * instantiated for the HOST over packed images made from the two GDBs: the check of that source against the oracle
 * where there is no GPU (tests/test_gap_core.py); not called by the product.  traces: scripts of Compute_Trace_PTS,
 * rewritten in place, traces->resume allocated and filled; fcap / hcap: the scratch of one "lane" in cells. */
int  fga_gap_core_check(const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns, fga_traces *traces,
                        int fcap, int64_t hcap);

/* ---- PAF output: replaces the ALNtoPAF process behind `FastGA -paf[m|x|s|S]` (ALNtoPAF.c:103-636; options 662-680).
 *      One line per alignment in set order.  With a CIGAR or cs tag the edit scripts of fga_trace_pts are needed; they
 *      are first regrouped into fewer, longer gaps exactly as Gap_Improver does (align.h:393-399, align.c:6714-7133;
 *      the reference's readers always call it right after Compute_Trace_PTS), on the host, one alignment per task.
 *      path NULL or "-": stdout. */
enum { FGA_PAF_CIGAR_M = 1,    /* -m  cg:Z: with M                                */
       FGA_PAF_CIGAR_X = 2,    /* -x  cg:Z: with = and X                          */
       FGA_PAF_CS_SHORT = 4,   /* -s  cs:Z: short form                            */
       FGA_PAF_CS_LONG = 8,    /* -S  cs:Z: long form                             */
       FGA_PAF_SWAP = 16,      /* -w  genome 2 as the query                       */
       FGA_OUT_PSL = 32 };     /* fga_run_params.paf_flags only: PSL lines (fga_write_psl) instead of PAF */
int  fga_write_paf(const char *path, const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns,
                   const fga_traces *traces /* NULL without CIGAR / cs */, int flags, int nthreads);
/* ---- reading a .1aln back: replaces open_Aln_Read + Read_Aln_Overlap + Read_Aln_Trace (alncode.c:62-237) for a whole
 *      file -- the reference's own binary files (list codecs included), ours, or the text form (ours, or what ONEview
 *      prints; the reference's seeking readers refuse text).  db1 / db2 receive the GDB paths of the
 *      file's reference lines, resolved against its directory line (db2 NULL for a self comparison); free() them.
 *      With fga_trace_pts + fga_write_paf / fga_write_psl this is the in-process ALNtoPAF / ALNtoPSL
 *      (fastga_amd/bin/ALNtoPAF, ALNtoPSL). */
int  fga_read_1aln(const char *path, fga_alns **out, int *tspace, char **db1, char **db2);

/* PSL output: replaces the ALNtoPSL process behind `FastGA -psl` (ALNtoPSL.c:77-405); always needs the edit scripts */
int  fga_write_psl(const char *path, const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns,
                   const fga_traces *traces, int nthreads);
/* Gap_Improver alone, in place on a whole set (trace and diffs as the reference leaves them in Path) */
int  fga_gap_improve(const fga_gdb *g1, const fga_gdb *g2 /* NULL: self */, const fga_alns *alns, fga_traces *traces);

/* ---- exact-signature shims of the reference's module seams ------------------------------------------------
 *      The same prototypes as the reference's cross-file functions, so that ONE call inside the unmodified pipeline can
 *      be pointed at the device (or A/B-ed against align.c / RSDsort.c): argument meaning, result fields, ownership (the
 *      trace lives in the Work_Data and is overwritten by the next call) and the 1-on-failure convention are the
 *      reference's.  Opaque arguments are `void *` exactly as align.h declares Work_Data and Align_Spec; `align` points at
 *      an Alignment (align.h:145-152) whose path is a Path (align.h:89-95); `range` at Range[nthreads] (FastGA.c:143-147).
 *      Parity devices (each call uploads its inputs), not the fast path.                                              */
void *fga_shim_New_Work_Data(void);                                                   /* align.h:166  New_Work_Data   */
void  fga_shim_Free_Work_Data(void *work);                                            /* align.h:168  Free_Work_Data  */
void *fga_shim_New_Align_Spec(double ave_corr, int trace_space, float *freq, int reach);   /* align.h:196 New_Align_Spec  */
void  fga_shim_Free_Align_Spec(void *spec);                                           /* align.h:198  Free_Align_Spec */
int   fga_shim_Local_Alignment(void *align, void *work, void *spec,                   /* align.h:235-236              */
                               int low, int hgh, int anti, int lbord, int hbord);
int   fga_shim_rmsd_sort(uint8_t *array, int64_t nelem, int rsize, int ksize,         /* FastGA.c:149-150, RSDsort.c:292 */
                         int nparts, int64_t *part, int nthreads, void *range);
/*      The pair of calls every reader of a .1aln makes per record (ALNtoPAF.c:278-280, ALNtoPSL.c:193-197): the trace
 *      points of align->path become the edit script (ints in the Work_Data, path->trace / tlen / diffs as the reference
 *      leaves them), then the script is regrouped in place.  Trace spacing 100, mode GREEDIEST (0), dlow > dhgh -- what the
 *      reference's own callers pass; `work` is a fga_shim_New_Work_Data packet.                                            */
int   fga_shim_Compute_Trace_PTS(void *align, void *work, int trace_spacing,              /* align.h:266-267, align.c:6171 */
                                 int mode, int dlow, int dhgh);
int   fga_shim_Gap_Improver(void *align, void *work);                                     /* align.h:399, align.c:6714     */

/* ---- the whole hot path: what `FastGA -1:<out> <root1> [<root2>]` does between "GIX present" and ".1aln closed" */
typedef struct
  { int     device;
    int     freq;            /* -f (10)                    */
    int     soft_mask;       /* -M or #mask                */
    int     symmetric;       /* -S                         */
    int     chain_break;     /* 2 * -s (2000)              */
    int     chain_min;       /* 2 * -c (170)               */
    int     align_min;       /* -l (100)                   */
    double  align_rate;      /* 1 - (-i) (0.3)             */
    int     nthreads;        /* host threads (-T)          */
    const char *out_path;    /* <name>.1aln (NULL: no file) */
    const char *command_line;
    const char *paf_path;    /* PAF output after the .1aln ("-": stdout, NULL: none)      */
    int     paf_flags;       /* FGA_PAF_* (-pafm / -pafx / -pafs / -pafS)                 */
    int64_t pass_seeds;      /* most seeds one sort / search pass takes (0: 1.5 G); more -> phase 2 runs over A-contig
                                parts, the reference's NPARTS loop (FastGA.c:5186-5204)  */
    int     build_index;     /* fga_run: build the genome indices on the device even when <root>.gix files exist */
    /* `#<mask>` arguments (FastGA.c:4568-4573): .1ano / .ano files whose union is the genome's soft mask ("" = the GDB's own
       lower-case mask); a genome with masks named gets its index built anew, like the reference's GIXmake call, and the
       comparison runs with soft masking on.  fga_run only. */
    const char *const *masks1; int nmasks1;
    const char *const *masks2; int nmasks2;
    int     reference_threads; /* n > 0: records that tie on (aread, abpos) in the order `FastGA -T<n>` writes them -- by the
                                slot of the search thread that held the A contig's panel of that strand (la_merge,
                                FastGA.c:3906-3918; fga_reference_slots); 0: by (bread, strand, survival)           */
  } fga_run_params;

typedef struct
  { int64_t nseeds, seed_len_sum, nhits, nunits, nalns, nlive, cover, ncalls, nwaves;
    double  load_s, upload_s, merge_s, sort_s, download_s, chain_s, extend_s, filter_s, write_s, phase23_s;
    double  trace_s, paf_s;  /* PAF output only: edit scripts on the device, regrouping + formatting on the host */
    float   merge_kernel_ms, sort_kernel_ms, extend_kernel_ms, trace_kernel_ms;
    int     nparts;          /* A-contig parts phase 2 was run over */
    int64_t bases1, bases2;  /* bases of the two genomes (the reference's "seeds per G1 position") */
    int64_t ext_cells, ext_bases, ext_trace;   /* extension accounting summed over the parts (fga_alns)         */
    double  ext_busy_waves;                    /* wavefronts busy on average (last part)                        */
    int64_t hbm_peak_bytes;                    /* peak device memory in use by this process, sampled at the stage boundaries */
    int64_t sort_keys;                         /* records sorted, summed over the parts; sort_passes radix passes each:      */
    int     sort_passes;                       /*   algorithmic traffic of the sort = 2 x 16 B x sort_keys x sort_passes     */
    int     streamed_parts;                    /* parts whose records went to the .1aln while later parts' kernels ran (0:   */
                                               /*   the file was written after the last part)                                */
    float   chain_kernel_ms;                   /* the chain scan's kernels (HIP events), summed over the parts               */
  } fga_run_stats;

int  fga_run(const char *root1, const char *root2 /* NULL: self */, const fga_run_params *prm, fga_run_stats *stats);

/* ---- one comparison cut into A-contig parts: several GPUs, or several passes of one --------------------
 *      Replaces the reference's partition glue -- Select[] / IDBsplit[] / NPARTS (FastGA.c:5057-5095), the seed file
 *      matrix N_Units / C_Units + buck[] counts and its transpose (FastGA.c:5097-5134, 5160-5184, 4160-4187) -- and the
 *      final merge of the per-thread record files (la_merge, FastGA.c:3991-4133; here fga_session_finish).
 *      Phase 1 shards by 12-mer prefix range (fga_merge_params.prefix_begin/end, the reference's thread split
 *      FastGA.c:2291-2321), phase 2 by A-contig part: contig pairs are independent work units, the redundancy filter
 *      needs all records of a contig pair together, and those all come from the part that owns the A contig.          */
/* seeds per A contig (length-sorted index) of a seed buffer: the reference's buck[] counts */
int  fga_seeds_contig_histogram(fga_dev *dev, const fga_dseeds *seeds, int nctg, int64_t *counts /* host, nctg */);
/* the same per strand: counts[u*nctg + j], u = 0 (N stream) / 1 (C stream): the buck[] of the reference's N_Units / C_Units */
int  fga_seeds_strand_histogram(fga_dev *dev, const fga_dseeds *seeds, int nctg, int64_t *counts /* host, 2*nctg */);
/* ---- the reference's order of records that tie on (aread, abpos): replaces what la_sort + la_merge (FastGA.c:3800-3835,
 *      3906-3918) make of the search threads' Range[] (rmsd_sort, RSDsort.c:318-343; FastGA.c:4336-4345) -- fga_order.c.
 *      fga_rmsd_ranges     the panel ranges rmsd_sort cuts for `nthreads` threads (part[x] = bytes of panel x)
 *      fga_reference_slots slot[u*nctg + j] = search thread of (strand u, A contig j) in `FastGA -T<nthreads>`, from the
 *                          per-strand seed counts, the A contig lengths in index order and the sort record width
 *      fga_alns_reference_order  a filtered set (aread, abpos, bread, comp, survival) -> (aread, abpos, slot, bread, ..) */
int  fga_rmsd_ranges(const int64_t *part, int nparts, int64_t asize, int nthreads, int *beg, int *end, int64_t *off);
int  fga_reference_slots(const int64_t *counts /* 2*nctg */, const int64_t *clen /* nctg */, int nctg, int nthreads, int swide,
                         int *slot /* 2*nctg */);
int  fga_alns_reference_order(fga_alns *alns, const int *slot, const int *invp /* original contig -> index order */, int nctg);
/* select[c] = part of A contig c: heaviest contig first, each to the lightest part so far.  A pure function of its
   arguments (every rank computes the same map from the all-reduced counts) */
int  fga_partition_contigs(const int64_t *weight, int nctg, int nparts, int *select);
/* the same contigs dealt out contiguously in their ORIGINAL order (perm[j] = original index of contig j), stretches of about
   equal weight: the records of part p then all come before those of part p+1 in the .1aln (fga_session_run writes a part's
   stretch while the next part's kernels run) */
int  fga_partition_contigs_in_order(const int64_t *weight, const int *perm, int nctg, int nparts, int *select);
/* the seeds regrouped by part into a caller-provided DEVICE buffer of fga_seeds_count x 16 bytes (e.g. the send buffer
   of an all-to-all-v); part p occupies records [part_off[p], part_off[p+1]) */
int  fga_seeds_split_to(fga_dev *dev, const fga_dseeds *seeds, const int *select, int nctg, int nparts /* <= 64 */,
                        void *dst_device, int64_t *part_off /* host, nparts+1 */);
/* a seed buffer made of `npieces` runs of 16-byte records that already are in DEVICE memory (the receive buffer) */
int  fga_seeds_import(fga_dev *dev, const void *const *src_device, const int64_t *counts, int npieces, fga_dseeds **out);
/* one dense stretch of device memory the caller keeps (a part's piece of its own fga_seeds_split_to buffer) as a seed buffer:
   nothing copied, fga_seeds_free leaves the stretch alone */
int  fga_seeds_view(fga_dev *dev, const void *src_device, int64_t count, fga_dseeds **out);
const void *fga_seeds_device_ptr(const fga_dseeds *seeds);
/* 12-mer prefix ranges [cuts[r], cuts[r+1]) of equal merge cost: the phase-1 shards (FastGA.c:2291-2321) */
int  fga_merge_prefix_cuts(fga_dev *dev, const fga_dgix *t1, const fga_dgix *t2 /* NULL: self */, int nshards,
                           int64_t *cuts /* host, nshards+1 */);
/* the record sets of several parts as one set in (part, unit, seq) discovery order, unit numbers made distinct: what
   the gathering rank hands to the redundancy filter (host only) */
int  fga_alns_concat(const fga_alns *const *raw, int nraw, fga_alns **out);

/* filtered sets (outputs of fga_filter_alignments[_mt], one per A-contig part) as one set in final order: their runs per
   A contig laid out by contig -- the reference's la_merge over its per-thread files (FastGA.c:3991-4133); host only */
int  fga_alns_merge_filtered(const fga_alns *const *filtered, int nsets, fga_alns **out);
int     fga_alns_merge_filtered_mt(const fga_alns *const *filtered, int nsets, int nthreads, fga_alns **out);   /* the copies on all threads */

/* pieces that lie in the memory of OTHER devices of the node (src_device_id[k] = HIP device of piece k): hipMemcpyPeerAsync
   over xGMI instead of the reference's re-read of the seed files of one part (FastGA.c:5160-5184, 4160-4187) */
int  fga_seeds_import_peer(fga_dev *dev, const void *const *src_device, const int *src_device_id, const int64_t *counts,
                           int npieces, fga_dseeds **out);
int  fga_dev_enable_peer(fga_dev *dev, int peer_device);   /* direct xGMI access dev -> peer (no-op when there is no path) */
int  fga_dev_device_count(void);                           /* HIP devices this process sees (0: none)                    */

/* ---- ONE comparison over `ndev` GPUs of one node from ONE process: the reference's parts machinery -- Select[] / IDBsplit[]
 *      and the unit matrix (FastGA.c:5057-5134), the transpose + NPARTS loop (FastGA.c:5160-5204), la_merge
 *      (FastGA.c:3991-4133) -- laid over the devices: one host thread + HIP stream per device, rank r merges its 12-mer
 *      prefix range on devices[r], the seeds move by A-contig part with hipMemcpyPeerAsync, rank p runs phase 2 and the
 *      redundancy filter on its part, the calling thread writes ONE .1aln (PAF / PSL) in the reference's order.  The
 *      result does not depend on ndev; devices may repeat ({0,0}: two ranks on GPU 0).  ndev = 1 is fga_run on devices[0].
 *      prm->device is ignored; prm->nthreads is shared out among the ranks.  fastga_amd/bin/FastGA: -G<n> / FGA_DEVICES. */
int  fga_run_multi(const char *root1, const char *root2 /* NULL: self */, const fga_run_params *prm,
                   int ndev, const int *devices, fga_run_stats *stats);
/*      The same as a session, the way the reference keeps its parts machinery for the length of its process: open reads the
 *      inputs once, puts every rank's slice on its device, enables peer access and leaves one host thread per device waiting;
 *      every run is one comparison from resident inputs (fga_run_multi = open + run + close); the contigs are dealt to the
 *      ranks by the wave steps their units took in the session's previous run (the first run: by seed counts) -- the result
 *      does not depend on it.  prm of open: nthreads, build_index, masks; prm of run: everything else.  One run at a time. */
typedef struct fga_multi fga_multi;
int  fga_multi_open(const char *root1, const char *root2 /* NULL: self */, const fga_run_params *prm,
                    int ndev, const int *devices, fga_multi **out);
int  fga_multi_run(fga_multi *m, const fga_run_params *prm, fga_run_stats *stats);
void fga_multi_close(fga_multi *m);
int  fga_multi_ndev(const fga_multi *m);
/* N1 E1 + N2 E2 over all ranks' slices, the reference's seed record width, the bases of the two genomes */
int  fga_multi_info(const fga_multi *m, int64_t *table_bytes, int *seed_bytes, int64_t *bases1, int64_t *bases2);
/* figures of one rank in the last run: seconds of phase 1 / exchange / phase 2 (with its filter), its extension kernel's
   ms and wave steps */
int  fga_multi_rank_stats(const fga_multi *m, int rank, double *seconds3, double *extend_kernel_ms, int64_t *wave_steps);

/* the same with the inputs kept resident in HBM between passes (what bench.py times) */
typedef struct fga_session fga_session;
int      fga_session_open(const char *root1, const char *root2, int device, fga_session **out);
/* nthreads = the -T the reference would hand to the GIXmake it runs for a missing index (fga_session_open: 8) */
int      fga_session_open_threads(const char *root1, const char *root2, int device, int nthreads, fga_session **out);
enum { FGA_SESSION_BUILD_INDEX = 1 };   /* indices built on the device even when <root>.gix files exist */
int      fga_session_open_flags(const char *root1, const char *root2, int device, int nthreads, int flags, fga_session **out);
/* the same with mask files named for the genomes (see fga_run_params.masks1): their indices are built on the device */
int      fga_session_open_masked(const char *root1, const char *root2, int device, int nthreads, int flags,
                                 const char *const *masks1, int nmasks1, const char *const *masks2, int nmasks2,
                                 fga_session **out);
/* the soft mask of an opened GDB becomes the union of the masks named (Read_ANO + ANO_Union, ANO.c:105-512, 678-830) */
int      fga_gdb_apply_masks(fga_gdb *gdb, const char *const *paths, int npaths);
/* rank `rank` of `nranks` of one comparison: the session holds only ITS 12-mer prefix range of both tables (the genomes'
 * bases stay whole: phase 2 needs them).  The ranges are cut for equal merge cost from the tables' per-prefix counts, the
 * same on every rank without communication; fga_session_prefix_cuts(s, nranks, ..) returns them, and fga_session_merge
 * takes the rank's own range. */
int      fga_session_open_sliced(const char *root1, const char *root2, int device, int nthreads, int rank, int nranks,
                                 fga_session **out);
int      fga_session_run(fga_session *s, const fga_run_params *prm, fga_run_stats *stats);
void     fga_session_close(fga_session *s);
fga_dev *fga_session_device(fga_session *s);
int64_t  fga_session_table_bytes(const fga_session *s);   /* N1*E1 + N2*E2                          */
int      fga_session_seed_bytes(const fga_session *s);    /* 1 + IBYTE + JBYTE of the reference seed */
int64_t  fga_session_bases(const fga_session *s, int which);
int      fga_session_nctg(const fga_session *s);          /* A contigs of the index (the partition's domain) */
const int *fga_session_contig_perm(const fga_session *s);  /* [nctg] original index of A contig j of the index order  */
int      fga_session_prefix_cuts(fga_session *s, int nshards, int64_t *cuts /* nshards+1 */);
/* per-strand seed counts of the A contigs (2 * fga_session_nctg values) that fga_session_merge has accumulated when
   prm->reference_threads > 0; ranks of a sharded run add theirs up (all-reduce) and hand the sums to the rank that finishes */
int      fga_session_strand_counts(const fga_session *s, int64_t *counts);
int      fga_session_set_strand_counts(fga_session *s, const int64_t *counts);
void     fga_session_clear_strand_counts(fga_session *s);     /* before the merges of a new comparison (fga_session_run does) */
/* fga_session_run in its three stages (stats are accumulated into *stats, which the caller zeroes once):
 *   merge  : phase 1 over a 12-mer prefix range (0,0 = all)                      -> seeds in HBM
 *   align  : phase 2 on a set of seeds (consumed): sort, chain scan, extension   -> accepted alignments, unfiltered
 *   finish : redundancy filter + phase 3 over the record sets of all parts      -> .1aln (and PAF / PSL)          */
int      fga_session_merge(fga_session *s, const fga_run_params *prm, int64_t prefix_begin, int64_t prefix_end,
                           fga_dseeds **out, fga_run_stats *stats);
int      fga_session_align(fga_session *s, const fga_run_params *prm, fga_dseeds *seeds, fga_alns **raw,
                           fga_run_stats *stats);
int      fga_session_finish(fga_session *s, const fga_run_params *prm, const fga_alns *const *raw, int nraw,
                            fga_run_stats *stats);
/* finish for record sets that were filtered part by part already (fga_filter_alignments_mt on each part's raw set, e.g. on
   the rank that produced it): merge by A contig + phase 3 */
int      fga_session_finish_filtered(fga_session *s, const fga_run_params *prm, const fga_alns *const *filtered, int nsets,
                                     fga_run_stats *stats);
/* finish as a stream, for parts that are contiguous stretches of A contigs in their original order
   (fga_partition_contigs_in_order): the stream on the session's genomes (prm: out_path, command_line), and one part's filtered
   set put into the reference's tie order (reference_threads > 0; the session's strand counts must be those of ALL prefix
   ranges: fga_session_set_strand_counts) -- then fga_aln_stream_append part by part, fga_aln_stream_close.  What
   fga_session_run does with its passes and fga_multi_run with its ranks; replaces la_merge (FastGA.c:3991-4133) */
int      fga_session_stream_open(fga_session *s, const fga_run_params *prm, fga_aln_stream **out);
int      fga_session_reference_order(fga_session *s, const fga_run_params *prm, fga_alns *set);

#ifdef __cplusplus
}
#endif
#endif
