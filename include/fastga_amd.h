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
int            fga_gix_open(const char *path, fga_gix **out);
void           fga_gix_close(fga_gix *gix);
int64_t        fga_gix_nents(const fga_gix *gix);
int            fga_gix_ebytes(const fga_gix *gix);
int            fga_gix_postbytes(const fga_gix *gix);
int            fga_gix_contbytes(const fga_gix *gix);
int            fga_gix_nctg(const fga_gix *gix);
int            fga_gix_nparts(const fga_gix *gix);
int64_t        fga_gix_maxpre(const fga_gix *gix);
const int     *fga_gix_perm(const fga_gix *gix);
const int64_t *fga_gix_index(const fga_gix *gix);
const uint8_t *fga_gix_table(const fga_gix *gix);

#ifdef __cplusplus
}
#endif
#endif
