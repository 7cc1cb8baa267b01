/* fga_shard.c -- the A-contig partition of one comparison (host).
 *
 * Replaces the reference's partition glue (FastGA.c:5057-5095): there Select[] cuts the length-sorted A contigs into
 * NPARTS consecutive runs of at least NTHREADS contigs and seqtot/NTHREADS bases, so that the sort panels of one part
 * fit in memory and every search thread owns some contigs.  Here a part is what one GPU (or one pass of one GPU) takes
 * through sort -> chain scan -> extension -> record gather, contig pairs being independent work units, and the cut is
 * made on the per-contig SEED counts of the merge (what the xGMI exchange and the sort actually move), not on bases:
 * longest-processing-time-first assignment, heaviest contig first, each to the currently lightest part.  The result
 * depends only on (weight[], nparts), so every rank computes the same map without communicating it.
 */
#include <stdlib.h>
#include <string.h>

#include "fga_host.h"
#include "fastga_amd.h"

typedef struct { int64_t w; int c; } wc;

static int heavier_first(const void *l, const void *r)
{ const wc *a = l, *b = r;
  if (a->w != b->w) return a->w > b->w ? -1 : 1;
  return a->c - b->c;
}

int fga_partition_contigs(const int64_t *weight, int nctg, int nparts, int *select)
{ wc *order;
  int64_t *load;
  int i, p;
  if (weight == NULL || select == NULL || nctg < 0 || nparts < 1)
    { fga_set_error("fga_partition_contigs: bad argument");
      return 1;
    }
  order = malloc(sizeof(wc)*(nctg > 0 ? nctg : 1));
  load  = calloc(nparts,sizeof(int64_t));
  if (order == NULL || load == NULL)
    { free(order); free(load);
      fga_set_error("out of memory");
      return 1;
    }
  for (i = 0; i < nctg; i++)
    { order[i].w = weight[i] > 0 ? weight[i] : 0; order[i].c = i; }
  qsort(order,nctg,sizeof(wc),heavier_first);
  for (i = 0; i < nctg; i++)
    { int best = 0;
      for (p = 1; p < nparts; p++)
        if (load[p] < load[best])
          best = p;
      select[order[i].c] = best;
      load[best] += order[i].w + 1;        /* +1: contigs without seeds still spread over the parts */
    }
  free(order); free(load);
  return 0;
}

/* the same domain dealt out CONTIGUOUSLY in the contigs' original order (perm[j] = original index of contig j of the index
   order): part p holds a stretch of original contigs, the stretches of about equal weight -- so that the records of part p
   all come before those of part p+1 in a .1aln (whose primary order is the original A contig).  Contigs beyond the genome's
   own (an index pads a short genome to one contig per thread) go to the last part */
int fga_partition_contigs_in_order(const int64_t *weight, const int *perm, int nctg, int nparts, int *select)
{ int64_t total = 0, acc = 0;
  int *byorig;
  int i, p = 0, nmax = 0;
  if (weight == NULL || perm == NULL || select == NULL || nctg < 0 || nparts < 1)
    { fga_set_error("fga_partition_contigs_in_order: bad argument");
      return 1;
    }
  for (i = 0; i < nctg; i++)
    { total += weight[i] > 0 ? weight[i] : 0;
      if (perm[i] + 1 > nmax) nmax = perm[i] + 1;
    }
  byorig = malloc(sizeof(int)*(nmax > 0 ? nmax : 1));
  if (byorig == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  for (i = 0; i < nmax; i++) byorig[i] = -1;
  for (i = 0; i < nctg; i++)
    if (perm[i] >= 0) byorig[perm[i]] = i;
  for (i = 0; i < nmax; i++)
    { const int j = byorig[i];
      if (j < 0) continue;
      const int64_t w = weight[j] > 0 ? weight[j] : 0;
      /* the next part begins with the contig whose middle lies beyond this part's share of the total (one step per contig:
         no part without a contig while contigs are left) */
      if (p < nparts-1 && i > 0 && (acc + w/2) * nparts >= total * (int64_t) (p+1))
        p += 1;
      select[j] = p;
      acc += w;
    }
  free(byorig);
  return 0;
}
