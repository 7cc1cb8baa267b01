/* fga_gix.c -- genome index (GIX) loader and plain-C producer.
 *
 * On-disk format (reference GIXmake.c:1490-1580, 1211-1278; libfastk.c:785-907):
 *   <root>.gix        int32 kmer(40) nparts minval(1) ibyte(3); int64 index[2^24] inclusive cumulative entry
 *                     counts per 12-mer prefix; int32 PostBytes ContBytes nparts; int64 maxpre; int32 freq(0)
 *                     ncontig; int32 Perm[ncontig]; int64 -1
 *   .<root>.ktab.<p>  int32 kmer; int64 nents; nents entries of E = 9+PostBytes+ContBytes bytes:
 *                     [0..6] bases 13..40 (4 per byte, first base in the high bits), [7] soft-mask length,
 *                     [8] lcp with the previous entry in bases (40 for duplicates, 0 for the first entry of a
 *                     part), [9..) little-endian in-contig position, then little-endian length-sorted contig
 *                     index with bit 7 of the last byte = complement strand.
 * Indexed k-mers: every 40-mer that *starts* with a closed (12,8)-syncmer under the 4-mer code order TMap,
 * in both orientations (GIXmake.c:37-39, 92-109, 406-611):  the 12-mer at j is a syncmer iff the minimum
 * canonical 8-mer code over offsets j..j+4 is attained at j or at j+4.  Forward entry: k-mer [j,j+40),
 * position j, needs j <= len-40.  Complement entry: reverse complement of [j-28,j+12), position j+12 (one
 * past the forward extent; `bost += TMER`, GIXmake.c:929-941), needs j >= 28.
 *
 * The producer below is a data-parallel restatement (count -> scatter -> per-panel sort), not the
 * reference's distribute / re-import / MSD-sort pipeline.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>

#include "fga_host.h"

/* The 4-mer code order that defines which 8-mers are "small" (format constant, GIXmake.c:92-109). */
static const uint8_t TMap[256] =
 { 0xff, 0xd4, 0xf5, 0xfd, 0xe4, 0xad, 0x21, 0xa5, 0xed, 0x64, 0xbf, 0xa9, 0xf3, 0x70, 0xd6, 0xf0,
   0xca, 0x89, 0xcb, 0xc9, 0x82, 0x9d, 0x13, 0x79, 0x0a, 0x0f, 0x25, 0x19, 0x3e, 0x47, 0xa3, 0xa8,
   0xf9, 0x5e, 0xe8, 0xa1, 0xb0, 0x71, 0x1d, 0x8c, 0xde, 0x69, 0xe7, 0x7c, 0x56, 0x3f, 0x90, 0xa4,
   0xeb, 0x45, 0x59, 0xf1, 0x97, 0x4c, 0x08, 0xa0, 0xb8, 0x4a, 0x86, 0xc8, 0xcd, 0x98, 0x7d, 0xfc,
   0xef, 0x4d, 0x83, 0x7e, 0xdc, 0x66, 0x2b, 0x8e, 0xe0, 0xa7, 0xd0, 0xa2, 0x88, 0x5f, 0x7f, 0xd9,
   0x9b, 0x78, 0xd1, 0x8b, 0xc3, 0x8f, 0x2d, 0xe6, 0x18, 0x27, 0x2c, 0x24, 0x94, 0xb7, 0xce, 0xbd,
   0x0d, 0x04, 0x1c, 0x09, 0x16, 0x23, 0x00, 0x1e, 0x1a, 0x29, 0x2e, 0x15, 0x01, 0x10, 0x2a, 0x20,
   0xbe, 0x31, 0x43, 0x58, 0xc2, 0xaa, 0x1f, 0xe5, 0xc5, 0x9e, 0xcf, 0xc6, 0x68, 0xb2, 0x80, 0xf4,
   0xf8, 0x53, 0xb6, 0x93, 0x76, 0x37, 0x11, 0x40, 0xda, 0x51, 0xba, 0x46, 0x42, 0x30, 0x60, 0x6d,
   0x5c, 0x39, 0x9f, 0x48, 0x6c, 0x62, 0x28, 0x67, 0x06, 0x12, 0x26, 0x0e, 0x33, 0x50, 0xa6, 0x63,
   0xdd, 0x3b, 0xab, 0x4b, 0x72, 0x5b, 0x22, 0x6f, 0xb4, 0x61, 0x92, 0x99, 0x36, 0x38, 0x65, 0xac,
   0x4f, 0x2f, 0x32, 0x44, 0x54, 0x3c, 0x03, 0x5d, 0x73, 0x3a, 0x77, 0x84, 0x8d, 0x4e, 0x49, 0xd2,
   0xfb, 0x91, 0x6a, 0xcc, 0x8a, 0x35, 0x02, 0x55, 0x7a, 0x34, 0x96, 0x3d, 0xd3, 0x41, 0x85, 0xf2,
   0xb1, 0x75, 0xc4, 0xb5, 0xbb, 0xb3, 0x1b, 0xd5, 0x07, 0x05, 0x17, 0x0b, 0x7b, 0xd7, 0xdf, 0xea,
   0xe3, 0x57, 0xc0, 0x95, 0x9c, 0x6e, 0x14, 0xae, 0xb9, 0x6b, 0xc1, 0x81, 0x87, 0x74, 0xd8, 0xe2,
   0xec, 0x52, 0xbc, 0xe9, 0xe1, 0xdb, 0x0c, 0xf7, 0xaf, 0x5a, 0x9a, 0xc7, 0xfa, 0xf6, 0xee, 0xfe
 };

/* ================================================================================================
 *  LOADER
 * ================================================================================================ */

static int read_full(int fd, void *buf, int64_t n)
{ uint8_t *p = buf;
  while (n > 0)
    { ssize_t x = read(fd,p,n > 0x40000000 ? 0x40000000 : n);
      if (x <= 0)
        return 1;
      p += x;
      n -= x;
    }
  return 0;
}

/* The pre-v1.3 index layout (what the reference still reads through Open_Post_List / old_merge_thread, FastGA.c:206-570,
 * 1027-1540, and what its EXAMPLE/sample_session shows): the .ktab.N parts hold every DISTINCT k-mer once -- the same
 * 9 leading bytes as today's entries, [0..6] bases 13..40, [7] the number of positions of the k-mer (today: the soft-mask
 * length), [8] lcp with the k-mer before -- and the positions sit, in table order, in .post.N parts (header: post
 * bytes, contig bytes, count).  The stub's 2^24 index counts table entries and is followed by a 2^16 index of
 * positions.  It is turned into today's in-memory table here: a k-mer with c positions becomes c entries, the first
 * with the stored lcp, the others with lcp 40, mask byte 0 (there was no soft masking), and the prefix index is
 * recounted -- the adaptive merge over the expanded table is the old merge over the counted one (same k-mer sets, same
 * run sizes in positions, same pairs).  X->freq keeps the cutoff the index was built with (k-mers above it are not
 * in an old table; the reference refuses -f above it, FastGA.c:4959-4974). */
static int open_legacy(fga_gix *X, const char *dir, const char *root, int npost)
{ const int pb = X->postbytes + X->contbytes;
  const int64_t nk = X->index[FGA_NPREFIX-1];
  uint8_t *old = NULL, *posts = NULL;
  int64_t *kpart = NULL;               /* first table entry of every .ktab part */
  char *name = NULL;
  int64_t off, np = 0, npos, e, o, pre, maxpre = 0;
  int f = -1, p;

  old = malloc((size_t) (nk > 0 ? nk : 1)*9);
  kpart = malloc(sizeof(int64_t)*(X->nparts+1));
  if (old == NULL || kpart == NULL) goto oom;
  off = 0;
  for (p = 1; p <= X->nparts; p++)
    { int32_t k;
      int64_t n;
      free(name);
      if (asprintf(&name,"%s/.%s.ktab.%d",dir,root,p) < 0) { name = NULL; goto oom; }
      f = open(name,O_RDONLY);
      if (f < 0)
        { fga_set_error("table part %s is missing",name);
          goto fail;
        }
      if (read_full(f,&k,sizeof(int32_t)) || read_full(f,&n,sizeof(int64_t))) goto ioerr;
      if (k != X->kmer || n < 0 || off+n > nk)
        { fga_set_error("table part %s does not match its stub",name);
          goto fail;
        }
      kpart[p-1] = off;
      if (read_full(f,old + off*9,n*9)) goto ioerr;
      off += n;
      close(f); f = -1;
    }
  kpart[X->nparts] = off;
  if (off != nk)
    { fga_set_error("index %s/%s: parts hold %lld k-mers, stub says %lld",dir,root,(long long) off,(long long) nk);
      goto fail;
    }
  for (p = 1; p <= npost; p++)                       /* sizes first, then the data */
    { int32_t hb[2];
      int64_t n;
      free(name);
      if (asprintf(&name,"%s/.%s.post.%d",dir,root,p) < 0) { name = NULL; goto oom; }
      f = open(name,O_RDONLY);
      if (f < 0)
        { fga_set_error("position list part %s is missing",name);
          goto fail;
        }
      if (read_full(f,hb,sizeof(hb)) || read_full(f,&n,sizeof(int64_t))) goto ioerr;
      if (hb[0] + hb[1] != pb || n < 0)
        { fga_set_error("position list part %s does not match its stub",name);
          goto fail;
        }
      np += n;
      close(f); f = -1;
    }
  posts = malloc((size_t) (np > 0 ? np : 1)*pb);
  if (posts == NULL) goto oom;
  npos = 0;
  for (p = 1; p <= npost; p++)
    { int32_t hb[2];
      int64_t n;
      free(name);
      if (asprintf(&name,"%s/.%s.post.%d",dir,root,p) < 0) { name = NULL; goto oom; }
      f = open(name,O_RDONLY);
      if (f < 0 || read_full(f,hb,sizeof(hb)) || read_full(f,&n,sizeof(int64_t)) || npos+n > np ||
          read_full(f,posts + npos*pb,n*pb))
        goto ioerr;
      npos += n;
      close(f); f = -1;
    }
  { int64_t tot = 0;
    for (e = 0; e < nk; e++)
      tot += old[e*9+7];
    if (tot != np)
      { fga_set_error("index %s/%s: the k-mer counts add up to %lld positions, the position lists hold %lld",dir,root,
                      (long long) tot,(long long) np);
        goto fail;
      }
  }

  /* the stub's prefix index drives the expansion below: it must be a cumulative count over the nk k-mers */
  { int64_t prev = 0;
    for (pre = 0; pre < FGA_NPREFIX; pre++)
      { if (X->index[pre] < prev || X->index[pre] > nk)
          { fga_set_error("index %s/%s: the prefix index of the stub is not a cumulative count over its %lld k-mers",
                          dir,root,(long long) nk);
            goto fail;
          }
        prev = X->index[pre];
      }
    if (prev != nk)
      { fga_set_error("index %s/%s: the prefix index of the stub ends at %lld, the table parts hold %lld k-mers",dir,root,
                      (long long) prev,(long long) nk);
        goto fail;
      }
  }

  X->ebytes  = 9 + pb;
  X->nents   = np;
  X->table   = malloc((size_t) np*X->ebytes + 64);
  X->partbeg = malloc(sizeof(int64_t)*(X->nparts+1));
  if (X->table == NULL || X->partbeg == NULL) goto oom;
  memset(X->table + np*X->ebytes,0,64);
  e = 0; o = 0; p = 0;
  for (pre = 0; pre < FGA_NPREFIX; pre++)
    { const int64_t eend = X->index[pre], obeg = o;
      for ( ; e < eend; e++)
        { const uint8_t *k = old + e*9;
          int c = k[7], q;
          while (p <= X->nparts && kpart[p] == e)
            X->partbeg[p++] = o;
          for (q = 0; q < c; q++, o++)
            { uint8_t *t = X->table + o*X->ebytes;
              memcpy(t,k,7);
              t[7] = 0;
              t[8] = q == 0 ? k[8] : (uint8_t) X->kmer;
              memcpy(t+9,posts + o*pb,pb);
            }
        }
      X->index[pre] = o;
      if (o - obeg > maxpre) maxpre = o - obeg;
    }
  while (p <= X->nparts)
    X->partbeg[p++] = o;
  X->maxpre = maxpre;
  X->legacy = 1;
  free(old); free(posts); free(kpart); free(name);
  return 0;

ioerr:
  fga_set_error("IO error reading %s",name);
  goto fail;
oom:
  fga_set_error("out of memory loading index %s/%s",dir,root);
fail:
  if (f >= 0) close(f);
  free(old); free(posts); free(kpart); free(name);
  return 1;
}

int fga_gix_open(const char *path, fga_gix **out)
{ fga_gix *X;
  char *noext = NULL, *dir = NULL, *root = NULL, *name = NULL;
  int   f = -1, p;
  int32_t hdr[4], tail3[3], fq, nctg;
  int64_t sentinel, off;

  *out = NULL;
  X = calloc(1,sizeof(fga_gix));
  if (X == NULL)
    { fga_set_error("out of memory");
      return 1;
    }

  { size_t n = strlen(path);
    noext = strdup(path);
    if (n > 4 && strcmp(path+n-4,".gix") == 0)
      noext[n-4] = '\0';
    else if (n > 5 && strcmp(path+n-5,".1gdb") == 0)
      noext[n-5] = '\0';
    else if (n > 4 && strcmp(path+n-4,".gdb") == 0)
      noext[n-4] = '\0';
  }
  dir  = fga_path_dir(noext);
  root = fga_path_root(noext,NULL);

  if (asprintf(&name,"%s/%s.gix",dir,root) < 0) goto oom;
  f = open(name,O_RDONLY);
  if (f < 0)
    { fga_set_error("cannot open index stub %s",name);
      goto fail;
    }
  if (read_full(f,hdr,sizeof(hdr)))
    goto ioerr;
  X->kmer   = hdr[0];
  X->nparts = hdr[1];
  if (X->kmer != FGA_KMER || hdr[3] != 3)
    { fga_set_error("%s: only k=40 / 3-byte-prefix indices are supported (k=%d ibyte=%d)",
                    name,hdr[0],hdr[3]);
      goto fail;
    }
  X->index = malloc(sizeof(int64_t)*FGA_NPREFIX);
  if (X->index == NULL) goto oom;
  if (read_full(f,X->index,sizeof(int64_t)*FGA_NPREFIX)) goto ioerr;
  if (read_full(f,tail3,sizeof(tail3))) goto ioerr;
  X->postbytes = tail3[0];
  X->contbytes = tail3[1];
  if (read_full(f,&X->maxpre,sizeof(int64_t))) goto ioerr;
  if (read_full(f,&fq,sizeof(int32_t))) goto ioerr;
  if (read_full(f,&nctg,sizeof(int32_t))) goto ioerr;
  X->freq = fq;
  X->nctg = nctg;
  X->perm = malloc(sizeof(int)*(nctg > 0 ? nctg : 1));
  if (X->perm == NULL) goto oom;
  if (read_full(f,X->perm,sizeof(int)*nctg)) goto ioerr;
  if (read_full(f,&sentinel,sizeof(int64_t))) goto ioerr;
  if (sentinel >= 0)                    /* pre-v1.3 layout: distinct k-mers with counts + separate .post.N files */
    { close(f);
      f = -1;
      if (open_legacy(X,dir,root,tail3[2]))
        goto fail;
      free(noext); free(dir); free(root); free(name);
      *out = X;
      return 0;
    }
  close(f);
  f = -1;

  X->ebytes  = 9 + X->postbytes + X->contbytes;
  X->nents   = X->index[FGA_NPREFIX-1];
  X->table   = malloc(X->nents*X->ebytes + 64);
  X->partbeg = malloc(sizeof(int64_t)*(X->nparts+1));
  if (X->table == NULL || X->partbeg == NULL) goto oom;
  memset(X->table + X->nents*X->ebytes,0,64);

  off = 0;
  for (p = 1; p <= X->nparts; p++)
    { int32_t k;
      int64_t n;
      free(name);
      if (asprintf(&name,"%s/.%s.ktab.%d",dir,root,p) < 0) { name = NULL; goto oom; }
      f = open(name,O_RDONLY);
      if (f < 0)
        { fga_set_error("table part %s is missing",name);
          goto fail;
        }
      if (read_full(f,&k,sizeof(int32_t)) || read_full(f,&n,sizeof(int64_t))) goto ioerr;
      if (k != X->kmer || off+n > X->nents)
        { fga_set_error("table part %s does not match its stub",name);
          goto fail;
        }
      X->partbeg[p-1] = off;
      if (read_full(f,X->table + off*X->ebytes,n*X->ebytes)) goto ioerr;
      off += n;
      close(f);
      f = -1;
    }
  X->partbeg[X->nparts] = off;
  if (off != X->nents)
    { fga_set_error("index %s/%s: parts hold %lld entries, stub says %lld",dir,root,
                    (long long) off,(long long) X->nents);
      goto fail;
    }

  free(noext); free(dir); free(root); free(name);
  *out = X;
  return 0;

ioerr:
  fga_set_error("IO error reading %s",name);
  goto fail;
oom:
  fga_set_error("out of memory loading index %s",path);
fail:
  if (f >= 0) close(f);
  free(noext); free(dir); free(root); free(name);
  free(X->index); free(X->perm); free(X->table); free(X->partbeg);
  free(X);
  return 1;
}

void fga_gix_close(fga_gix *X)
{ if (X == NULL) return;
  free(X->index); free(X->perm); free(X->table); free(X->partbeg);
  free(X);
}

int64_t        fga_gix_nents(const fga_gix *X)     { return X->nents; }
int            fga_gix_ebytes(const fga_gix *X)    { return X->ebytes; }
int            fga_gix_postbytes(const fga_gix *X) { return X->postbytes; }
int            fga_gix_contbytes(const fga_gix *X) { return X->contbytes; }
int            fga_gix_nctg(const fga_gix *X)      { return X->nctg; }
int            fga_gix_nparts(const fga_gix *X)    { return X->nparts; }
int64_t        fga_gix_part_begin(const fga_gix *X, int p) { return (p < 0 || p > X->nparts || X->partbeg == NULL) ? -1 : X->partbeg[p]; }
int64_t        fga_gix_maxpre(const fga_gix *X)    { return X->maxpre; }
const int     *fga_gix_perm(const fga_gix *X)      { return X->perm; }
int            fga_gix_legacy_cutoff(const fga_gix *X) { return X->legacy ? X->freq : 0; }
const int64_t *fga_gix_index(const fga_gix *X)     { return X->index; }
const uint8_t *fga_gix_table(const fga_gix *X)     { return X->table; }

/* ================================================================================================
 *  PRODUCER
 * ================================================================================================ */

typedef struct
  { uint64_t suf;     /* bases 13..40 in the low 56 bits, first base highest                    */
    uint64_t pay;     /* post | (contig|sign) << (8*postbytes), i.e. the payload as one LE int  */
    uint32_t pre;     /* 24-bit prefix (bases 1..12)                                            */
    uint8_t  mask;
  } krec;

typedef struct
  { const fga_gdb *gdb;
    const int     *invp;       /* original contig -> length-sorted index                 */
    int            postbytes, contbytes;
    int            pass;       /* 0 = count, 1 = scatter                                 */
    uint32_t      *count;      /* [2^24] per-prefix counts (pass 0, atomic)              */
    int64_t       *cursor;     /* [2^24] per-prefix write cursors (pass 1, atomic)       */
    krec          *recs;
    int            use_mask;   /* fill the soft-mask byte from gdb->mbeg/mend             */
    int           *next;       /* shared work counter over contigs                       */
    int64_t        nfwd, ncmp;
    int64_t       *sbuck;      /* [1024] shared: 5-base bucket counts of EVERY syncmer, both strands, whether or not
                                  its 40-mer fits the contig -- the reference sizes its table parts from this
                                  sample (sample_thread, GIXmake.c:163-327), not from the final table           */
  } scan_arg;

static inline uint8_t comp4(uint8_t x)     /* reverse complement of a packed 4-mer */
{ x = ~x;
  return (uint8_t) (((x & 0x03) << 6) | ((x & 0x0c) << 2) | ((x & 0x30) >> 2) | ((x & 0xc0) >> 6));
}

/* scan one contig; calls emit for every selected (position, strand) */
static void scan_contig(scan_arg *A, int c)
{ const fga_gdb *G = A->gdb;
  int64_t len = G->contigs[c].clen;
  uint8_t *seq;
  uint16_t *v8;
  int64_t j, p;
  uint64_t ctg, sign;

  if (len < 12)
    return;
  seq = malloc(len+2);
  v8  = malloc(sizeof(uint16_t)*(len+1));
  if (seq == NULL || v8 == NULL)
    { free(seq); free(v8);
      return;
    }
  fga_gdb_get_contig(G,c,seq);
  /* canonical code of the 8-mer at p, p in [0,len-8] */
  { uint8_t *s = seq+1;
    for (p = 0; p+8 <= len; p++)
      { uint8_t a = (uint8_t) ((s[p]<<6)|(s[p+1]<<4)|(s[p+2]<<2)|s[p+3]);
        uint8_t b = (uint8_t) ((s[p+4]<<6)|(s[p+5]<<4)|(s[p+6]<<2)|s[p+7]);
        uint16_t mn = (uint16_t) ((TMap[a]<<8) | TMap[b]);
        uint16_t mc = (uint16_t) ((TMap[comp4(b)]<<8) | TMap[comp4(a)]);
        v8[p] = mn < mc ? mn : mc;
      }
  }
  ctg  = (uint64_t) A->invp[c];
  sign = ((uint64_t) 0x80) << (8*(A->contbytes-1));
  /* soft mask byte of the k-mers whose syncmer starts at j: bases from j to the end of the mask interval that
     holds j, capped at 40; 0 outside intervals (setup_thread_with_masks, GIXmake.c:1100-1108)              */
  int64_t mi = 0, mtop = 0;
  if (A->use_mask && G->nmask > 0)
    { mi = G->moff[c]; mtop = G->moff[c+1]; }

  { uint8_t *s = seq+1;
    for (j = 0; j+12 <= len; j++)
      { uint16_t m = v8[j];
        int q;
        for (q = 1; q <= 4; q++)
          if (v8[j+q] < m) m = v8[j+q];
        if (v8[j] != m && v8[j+4] != m)
          continue;
        if (A->pass == 0)
          { uint32_t f5 = 0, c5 = 0;
            int k;
            for (k = 0; k < 5; k++)
              { f5 = (f5<<2) | s[j+k];
                c5 = (c5<<2) | (uint32_t) (3 - s[j+11-k]);
              }
            __atomic_fetch_add(A->sbuck+f5,1,__ATOMIC_RELAXED);
            __atomic_fetch_add(A->sbuck+c5,1,__ATOMIC_RELAXED);
          }
        uint8_t pbg = 0;
        if (mi < mtop)
          { while (mi < mtop && j >= G->mend[mi])
              mi += 1;
            if (mi < mtop && j >= G->mbeg[mi])
              { int64_t d = G->mend[mi] - j;
                pbg = (uint8_t) (d > FGA_KMER ? FGA_KMER : d);
              }
          }

        if (j <= len-FGA_KMER)                       /* forward k-mer [j,j+40) */
          { uint32_t pre = 0;
            int k;
            for (k = 0; k < 12; k++)
              pre = (pre<<2) | s[j+k];
            if (A->pass == 0)
              __atomic_fetch_add(A->count+pre,1,__ATOMIC_RELAXED);
            else
              { uint64_t suf = 0;
                int64_t  w;
                krec    *r;
                for (k = 12; k < 40; k++)
                  suf = (suf<<2) | s[j+k];
                w = __atomic_fetch_add(A->cursor+pre,1,__ATOMIC_RELAXED);
                r = A->recs + w;
                r->suf = suf; r->pre = pre; r->mask = pbg;
                r->pay = (uint64_t) j | (ctg << (8*A->postbytes));
              }
            A->nfwd += 1;
          }
        if (j >= FGA_KMER-12)                        /* complement k-mer = revcomp of [j-28,j+12) */
          { uint32_t pre = 0;
            int k;
            for (k = 0; k < 12; k++)
              pre = (pre<<2) | (3 - s[j+11-k]);
            if (A->pass == 0)
              __atomic_fetch_add(A->count+pre,1,__ATOMIC_RELAXED);
            else
              { uint64_t suf = 0;
                int64_t  w;
                krec    *r;
                for (k = 12; k < 40; k++)
                  suf = (suf<<2) | (3 - s[j+11-k]);
                w = __atomic_fetch_add(A->cursor+pre,1,__ATOMIC_RELAXED);
                r = A->recs + w;
                r->suf = suf; r->pre = pre; r->mask = pbg;
                r->pay = (uint64_t) (j+12) | ((ctg|sign) << (8*A->postbytes));
              }
            A->ncmp += 1;
          }
      }
  }
  free(seq);
  free(v8);
}

static void *scan_thread(void *arg)
{ scan_arg *A = arg;
  int n = A->gdb->ncontig;
  while (1)
    { int c = __atomic_fetch_add(A->next,1,__ATOMIC_RELAXED);
      if (c >= n)
        break;
      scan_contig(A,c);
    }
  return NULL;
}

static int krec_cmp(const void *l, const void *r)
{ const krec *a = l, *b = r;
  if (a->suf != b->suf) return a->suf < b->suf ? -1 : 1;
  if (a->pay != b->pay) return a->pay < b->pay ? -1 : 1;
  return 0;
}

typedef struct
  { krec          *recs;
    const int64_t *index;     /* inclusive cumulative */
    int           *next;
  } sort_arg;

#define SORT_CHUNK 4096       /* prefixes handed out per grab */

static void *sort_thread(void *arg)
{ sort_arg *S = arg;
  while (1)
    { int b = __atomic_fetch_add(S->next,SORT_CHUNK,__ATOMIC_RELAXED);
      int e, p;
      if (b >= FGA_NPREFIX)
        break;
      e = b + SORT_CHUNK;
      for (p = b; p < e; p++)
        { int64_t lo = (p == 0) ? 0 : S->index[p-1];
          int64_t n  = S->index[p] - lo;
          if (n > 1)
            { krec *r = S->recs + lo;
              if (n <= 16)
                { int64_t i, k;
                  for (i = 1; i < n; i++)
                    { krec t = r[i];
                      for (k = i-1; k >= 0 && krec_cmp(r+k,&t) > 0; k--)
                        r[k+1] = r[k];
                      r[k+1] = t;
                    }
                }
              else
                qsort(r,n,sizeof(krec),krec_cmp);
            }
        }
    }
  return NULL;
}

static int lcp80(uint32_t prea, uint64_t sufa, uint32_t preb, uint64_t sufb)
{ if (prea != preb)
    return (__builtin_clz(prea ^ preb) - 8) >> 1;
  if (sufa != sufb)
    return 12 + ((__builtin_clzll(sufa ^ sufb) - 8) >> 1);
  return 40;
}

static int write_full(int fd, const void *buf, int64_t n)
{ const uint8_t *p = buf;
  while (n > 0)
    { ssize_t x = write(fd,p,n > 0x40000000 ? 0x40000000 : n);
      if (x < 0)
        return 1;
      p += x;
      n -= x;
    }
  return 0;
}

/* Layout of the index of `G` for a given -T: contig order by decreasing length (ties keep the input order; the contig
 * count is padded to >= nthreads with fake 40-base contigs, short_GDB_fix, GIXmake.c:1605-1624), payload byte widths
 * and the number of table parts (GIXmake.c:1907-1963).  perm / invp are malloc'd. */
int fga_gix_layout(const fga_gdb *G, int nthreads, int *nctg_out, int **perm_out, int **invp_out,
                   int *postbytes_out, int *contbytes_out, int *nparts_out)
{ int      nreal = G->ncontig, nctg, i;
  int64_t *clen = NULL;
  int     *perm = NULL, *invp = NULL;
  int      postbytes, contbytes, nparts;

  if (nthreads < 1) nthreads = 1;
  if (nthreads > 32) nthreads = 32;
  nctg = nreal < nthreads ? nthreads : nreal;
  clen = malloc(sizeof(int64_t)*nctg);
  perm = malloc(sizeof(int)*nctg);
  invp = malloc(sizeof(int)*nctg);
  if (clen == NULL || perm == NULL || invp == NULL) goto oom;
  for (i = 0; i < nctg; i++)
    clen[i] = (i < nreal) ? G->contigs[i].clen : FGA_KMER;
  for (i = 0; i < nctg; i++)
    perm[i] = i;
  { int a;                            /* stable merge sort */
    int *tmp = malloc(sizeof(int)*nctg);
    int width;
    if (tmp == NULL) goto oom;
    for (width = 1; width < nctg; width *= 2)
      { for (a = 0; a < nctg; a += 2*width)
          { int mid = a+width < nctg ? a+width : nctg;
            int hi  = a+2*width < nctg ? a+2*width : nctg;
            int x = a, y = mid, k = a;
            while (x < mid && y < hi)
              tmp[k++] = (clen[perm[y]] > clen[perm[x]]) ? perm[y++] : perm[x++];
            while (x < mid) tmp[k++] = perm[x++];
            while (y < hi)  tmp[k++] = perm[y++];
          }
        memcpy(perm,tmp,sizeof(int)*nctg);
      }
    free(tmp);
  }
  for (i = 0; i < nctg; i++)
    invp[perm[i]] = i;

  { int64_t range = 0, cum;
    for (i = 0; i < nctg; i++)
      if (clen[i] > range) range = clen[i];
    postbytes = 0;
    for (cum = 1; cum < range; cum *= 256) postbytes += 1;
    range = 2*(int64_t) nctg;
    contbytes = 0;
    for (cum = 1; cum < range; cum *= 256) contbytes += 1;
  }
  if (postbytes + contbytes > 8)
    { fga_set_error("payload wider than 8 bytes is not supported");
      free(clen); free(perm); free(invp);
      return 1;
    }
  { int64_t seqtot = G->seqtot + (int64_t) (nctg-nreal)*FGA_KMER;
    int64_t nels = 0x100000000ll / (contbytes + postbytes + FGA_KMER/4 + 2);
    int     nbit = (int) ((.81 * (seqtot - (FGA_KMER-1)*(int64_t) nctg)) / nels);
    nparts = ((nbit-1)/nthreads+1)*nthreads;
    if (nparts < 8) nparts = 8;
    else if (nparts > 64) nparts = 64;
  }
  free(clen);
  *nctg_out = nctg; *perm_out = perm; *invp_out = invp;
  *postbytes_out = postbytes; *contbytes_out = contbytes; *nparts_out = nparts;
  return 0;
oom:
  fga_set_error("out of memory building index");
  free(clen); free(perm); free(invp);
  return 1;
}

/* split of the k-mer space into parts at 5-base (10-bit) buckets from the syncmer sample histogram (GIXmake.c:655-691) */
void fga_gix_ksplit(const int64_t *sbuck, int nparts, int *ksplit)
{ int64_t buck[1024], t;
  int b, n;
  buck[0] = sbuck[0];
  for (b = 1; b < 1024; b++)
    buck[b] = buck[b-1] + sbuck[b];
  ksplit[0] = 0;
  n = 1;
  t = buck[1023]/nparts;
  for (b = 0; b < 1024 && n < nparts; b++)
    if (buck[b] >= t)
      { int64_t prev = b > 0 ? buck[b-1] : 0;
        if (buck[b]-t > t-prev)
          ksplit[n] = b;
        else
          ksplit[n] = b+1;
        n += 1;
        t = (n*buck[1023])/nparts;
      }
  while (n <= nparts)
    ksplit[n++] = 1024;
  ksplit[nparts] = 1024;
}

const uint8_t *fga_gix_tmap(void) { return TMap; }

/* Write <root>.gix + .<root>.ktab.N from an index held in memory (index, table, partbeg, perm: e.g. the host copy of
 * fga_dgix_build) -- the same files fga_gix_build writes (GIXmake.c:1389-1600 for the layout). */
int fga_gix_write_files(const fga_gix *X, const char *target)
{ char *noext = NULL, *dir = NULL, *root = NULL, *name = NULL;
  int status = 1, part, fd = -1;
  if (X->index == NULL || X->table == NULL || X->partbeg == NULL || X->perm == NULL)
    { fga_set_error("fga_gix_write_files: the index has no host copy of its table");
      return 1;
    }
  noext = strdup(target);
  if (noext == NULL) goto oom;
  { size_t n = strlen(noext);
    if (n > 4 && strcmp(noext+n-4,".gix") == 0) noext[n-4] = '\0';
    else if (n > 4 && strcmp(noext+n-4,".gdb") == 0) noext[n-4] = '\0';
    else if (n > 5 && strcmp(noext+n-5,".1gdb") == 0) noext[n-5] = '\0';
  }
  dir  = fga_path_dir(noext);
  root = fga_path_root(noext,NULL);
  for (part = 0; part < X->nparts; part++)
    { int64_t lo = X->partbeg[part], hi = X->partbeg[part+1], nout = hi-lo;
      int32_t k = FGA_KMER;
      free(name); name = NULL;
      if (asprintf(&name,"%s/.%s.ktab.%d",dir,root,part+1) < 0) { name = NULL; goto oom; }
      fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0666);
      if (fd < 0)
        { fga_set_error("cannot open %s for writing",name);
          goto done;
        }
      if (write_full(fd,&k,sizeof(int32_t)) || write_full(fd,&nout,sizeof(int64_t)) ||
          write_full(fd,X->table + lo*X->ebytes,nout*X->ebytes))
        goto ioerr;
      close(fd); fd = -1;
    }
  { int32_t x;
    int64_t y;
    free(name); name = NULL;
    if (asprintf(&name,"%s/%s.gix",dir,root) < 0) { name = NULL; goto oom; }
    fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0666);
    if (fd < 0)
      { fga_set_error("cannot open %s for writing",name);
        goto done;
      }
    x = FGA_KMER;  if (write_full(fd,&x,4)) goto ioerr;
    x = X->nparts; if (write_full(fd,&x,4)) goto ioerr;
    x = 1;         if (write_full(fd,&x,4)) goto ioerr;
    x = 3;         if (write_full(fd,&x,4)) goto ioerr;
    if (write_full(fd,X->index,sizeof(int64_t)*FGA_NPREFIX)) goto ioerr;
    x = X->postbytes; if (write_full(fd,&x,4)) goto ioerr;
    x = X->contbytes; if (write_full(fd,&x,4)) goto ioerr;
    x = X->nparts;    if (write_full(fd,&x,4)) goto ioerr;
    y = X->maxpre;    if (write_full(fd,&y,8)) goto ioerr;
    x = 0;            if (write_full(fd,&x,4)) goto ioerr;
    x = X->nctg;      if (write_full(fd,&x,4)) goto ioerr;
    if (write_full(fd,X->perm,sizeof(int)*X->nctg)) goto ioerr;
    y = -1;           if (write_full(fd,&y,8)) goto ioerr;
    close(fd); fd = -1;
  }
  status = 0;
  goto done;
ioerr:
  fga_set_error("IO error writing %s",name);
  goto done;
oom:
  fga_set_error("out of memory");
done:
  if (fd >= 0) close(fd);
  free(noext); free(dir); free(root); free(name);
  return status;
}


/* Build <root>.gix + .<root>.ktab.* for `gdb`.  `nthreads` plays the role of GIXmake's -T: it sets the
 * worker count, the number of table parts (GIXmake.c:1907-1917) and the padding of the contig count to
 * >= nthreads with fake 40-base contigs (short_GDB_fix, GIXmake.c:1605-1624).                          */
int fga_gix_build_masked(const fga_gdb *G, const char *target, int nthreads, int use_mask);

int fga_gix_build(const fga_gdb *G, const char *target, int nthreads)
{ return fga_gix_build_masked(G,target,nthreads,0); }

/* use_mask: fill the per-entry soft-mask byte from the GDB's lower-case intervals (GIXmake's `#` argument) */
int fga_gix_build_masked(const fga_gdb *G, const char *target, int nthreads, int use_mask)
{ int      nreal = G->ncontig, nctg;
  int64_t *clen = NULL;
  int     *perm = NULL, *invp = NULL;
  int      postbytes, contbytes, ebytes, nparts;
  int64_t  sbuck[1024];
  uint32_t *count = NULL;
  int64_t  *index = NULL, *cursor = NULL;
  krec     *recs = NULL;
  int64_t   nents = 0, maxpre = 0;
  char *noext = NULL, *dir = NULL, *root = NULL, *name = NULL;
  uint8_t *obuf = NULL;
  int   status = 1, i;
  pthread_t *th = NULL;

  if (nthreads < 1) nthreads = 1;
  if (nthreads > 32) nthreads = 32;
  if (fga_gix_layout(G,nthreads,&nctg,&perm,&invp,&postbytes,&contbytes,&nparts))
    goto fail;
  ebytes = 9 + postbytes + contbytes;
  clen = malloc(sizeof(int64_t)*nctg);
  if (clen == NULL) goto oom;
  for (i = 0; i < nctg; i++)
    clen[i] = (i < nreal) ? G->contigs[i].clen : FGA_KMER;

  count = calloc(FGA_NPREFIX,sizeof(uint32_t));
  index = malloc(sizeof(int64_t)*FGA_NPREFIX);
  cursor = malloc(sizeof(int64_t)*FGA_NPREFIX);
  th = malloc(sizeof(pthread_t)*nthreads);
  if (count == NULL || index == NULL || cursor == NULL || th == NULL) goto oom;

  /* pass 0: count entries per 12-mer prefix; pass 1: scatter */
  memset(sbuck,0,sizeof(sbuck));
  { scan_arg *args = calloc(nthreads,sizeof(scan_arg));
    int pass, next;
    if (args == NULL) goto oom;
    for (pass = 0; pass < 2; pass++)
      { next = 0;
        for (i = 0; i < nthreads; i++)
          { args[i].gdb = G; args[i].invp = invp;
            args[i].postbytes = postbytes; args[i].contbytes = contbytes;
            args[i].pass = pass; args[i].count = count; args[i].cursor = cursor; args[i].recs = recs;
            args[i].use_mask = use_mask;
            args[i].next = &next; args[i].nfwd = args[i].ncmp = 0;
            args[i].sbuck = sbuck;
          }
        for (i = 1; i < nthreads; i++)
          pthread_create(th+i,NULL,scan_thread,args+i);
        scan_thread(args);
        for (i = 1; i < nthreads; i++)
          pthread_join(th[i],NULL);
        if (pass == 0)
          { int64_t cum = 0;
            int p;
            maxpre = 0;
            for (p = 0; p < FGA_NPREFIX; p++)
              { cursor[p] = cum;
                if ((int64_t) count[p] > maxpre) maxpre = count[p];
                cum += count[p];
                index[p] = cum;
              }
            nents = cum;
            recs = malloc(sizeof(krec)*(nents+1));
            if (recs == NULL)
              { free(args);
                goto oom;
              }
          }
      }
    free(args);
  }

  /* per-panel sort by (suffix, payload): deterministic whatever the thread interleaving */
  { sort_arg S;
    int next = 0;
    S.recs = recs; S.index = index; S.next = &next;
    for (i = 1; i < nthreads; i++)
      pthread_create(th+i,NULL,sort_thread,&S);
    sort_thread(&S);
    for (i = 1; i < nthreads; i++)
      pthread_join(th[i],NULL);
  }

  /* split of the k-mer space into parts at 5-base (10-bit) buckets (GIXmake.c:655-691) */
  noext = strdup(target);
  { size_t n = strlen(noext);
    if (n > 4 && strcmp(noext+n-4,".gix") == 0) noext[n-4] = '\0';
    else if (n > 4 && strcmp(noext+n-4,".gdb") == 0) noext[n-4] = '\0';
    else if (n > 5 && strcmp(noext+n-5,".1gdb") == 0) noext[n-5] = '\0';
  }
  dir  = fga_path_dir(noext);
  root = fga_path_root(noext,NULL);

  { int     ksplit[65];
    int     part;

    fga_gix_ksplit(sbuck,nparts,ksplit);

    obuf = malloc((size_t) ebytes * 65536);
    if (obuf == NULL) goto oom;

    for (part = 0; part < nparts; part++)
      { int64_t lo = ksplit[part] == 0 ? 0 : index[(ksplit[part]<<14)-1];
        int64_t hi = ksplit[part+1] == 0 ? 0 : index[(ksplit[part+1]<<14)-1];
        int64_t x, nout = hi-lo;
        int32_t k = FGA_KMER;
        int     fd, fill = 0;

        free(name);
        if (asprintf(&name,"%s/.%s.ktab.%d",dir,root,part+1) < 0) { name = NULL; goto oom; }
        fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0666);
        if (fd < 0)
          { fga_set_error("cannot open %s for writing",name);
            goto fail;
          }
        if (write_full(fd,&k,sizeof(int32_t)) || write_full(fd,&nout,sizeof(int64_t)))
          { close(fd); goto ioerr; }
        for (x = lo; x < hi; x++)
          { krec *r = recs+x;
            uint8_t *o = obuf + (size_t) fill*ebytes;
            int lcp, q;
            if (x == lo)
              lcp = 0;
            else
              lcp = lcp80(r[-1].pre,r[-1].suf,r->pre,r->suf);
            for (q = 0; q < 7; q++)
              o[q] = (uint8_t) (r->suf >> (8*(6-q)));
            o[7] = r->mask;
            o[8] = (uint8_t) lcp;
            for (q = 0; q < postbytes+contbytes; q++)
              o[9+q] = (uint8_t) (r->pay >> (8*q));
            if (++fill == 65536)
              { if (write_full(fd,obuf,(int64_t) fill*ebytes)) { close(fd); goto ioerr; }
                fill = 0;
              }
          }
        if (fill > 0 && write_full(fd,obuf,(int64_t) fill*ebytes)) { close(fd); goto ioerr; }
        close(fd);
      }
  }

  /* the stub */
  { int fd;
    int32_t x;
    int64_t y;
    free(name);
    if (asprintf(&name,"%s/%s.gix",dir,root) < 0) { name = NULL; goto oom; }
    fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0666);
    if (fd < 0)
      { fga_set_error("cannot open %s for writing",name);
        goto fail;
      }
    x = FGA_KMER; if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = nparts;   if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = 1;        if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = 3;        if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    if (write_full(fd,index,sizeof(int64_t)*FGA_NPREFIX)) { close(fd); goto ioerr; }
    x = postbytes; if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = contbytes; if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = nparts;    if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    if (write_full(fd,&maxpre,8)) { close(fd); goto ioerr; }
    x = 0;         if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    x = nctg;      if (write_full(fd,&x,4)) { close(fd); goto ioerr; }
    if (write_full(fd,perm,sizeof(int)*nctg)) { close(fd); goto ioerr; }
    y = -1;        if (write_full(fd,&y,8)) { close(fd); goto ioerr; }
    close(fd);
  }
  status = 0;
  goto done;

ioerr:
  fga_set_error("IO error writing %s",name);
  goto fail;
oom:
  fga_set_error("out of memory building index");
fail:
  status = 1;
done:
  free(clen); free(perm); free(invp); free(count); free(index); free(cursor); free(recs);
  free(noext); free(dir); free(root); free(name); free(obuf); free(th);
  return status;
}
