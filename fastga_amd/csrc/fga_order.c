/* fga_order.c -- the reference's order for records that tie on (aread, abpos) (host).
 *
 * The reference writes the records of search thread t to file t, sorts every file by (aread, abpos, bread, comp, file
 * position) (la_sort / SORT_MAP, FastGA.c:3800-3835) and merges the files with a heap whose comparison is (aread,
 * abpos, thread slot) (la_merge / MAPARE, FastGA.c:3906-3918).  The final order therefore is
 *
 *        (aread, abpos, slot, bread, comp, order of survival)
 *
 * where slot = the search thread that held the A contig's sort panel of that STRAND: for every A-contig part i
 * (IDBsplit[], FastGA.c:5057-5095) and strand u the panels of the part are cut into consecutive ranges of about
 * 1/NTHREADS of the part's seed bytes by rmsd_sort (RSDsort.c:318-343) and search thread t takes range t
 * (FastGA.c:4336-4345).  A pure function of the per-strand seed counts of the A contigs, the A contig lengths and -T:
 * independent of how many GPUs or passes produced the records.  fga_filter_alignments leaves (aread, abpos, bread, comp,
 * survival); the two differ only inside a run of equal (aread, abpos) that holds records of both strands whose slots
 * differ, where the strand of the smaller slot goes first.
 */
#include <stdlib.h>
#include <string.h>

#include "fga_host.h"
#include "fastga_amd.h"

/* the panel ranges rmsd_sort hands to its sort threads -- and, through Range[], to the search threads: consecutive
   panels of about asize/nthreads bytes each (RSDsort.c:318-343).  part[x] = bytes of panel x; returns the ranges in use */
int fga_rmsd_ranges(const int64_t *part, int nparts, int64_t asize, int nthreads, int *beg, int *end, int64_t *off)
{ int64_t thr = asize / nthreads, sum = 0, o = 0;
  int x = 0, n = 0, b;
  while (x < nparts && part[x] <= 0) x += 1;
  b = x;
  for (; x < nparts; x++)
    if (part[x] > 0)
      { sum += part[x];
        if (sum >= thr && n < nthreads)
          { beg[n] = b; end[n] = x+1; off[n] = o;
            n += 1;
            thr = (asize * (n+1)) / nthreads;
            b = x+1;
            o = sum;
          }
      }
  for (x = n; x < nthreads; x++)
    { beg[x] = end[x] = (n > 0 ? end[n-1] : 0); off[x] = asize; }
  return n;
}

/* slot[u*nctg + j] = search thread of strand u (0: N, 1: C) and A contig j (length-sorted index) in `FastGA -T<nthreads>`;
   -1 for a contig without seeds on that strand.
     counts[u*nctg + j]  seeds of strand u whose A contig is j (fga_seeds_strand_histogram, summed over all ranks)
     clen[j]             contig lengths in the index's order, fake contigs of a short GDB included (short_GDB_fix,
                         FastGA.c:4413-4432)
     swide               bytes of a sort record: 2*DBYTE + JCONT + 2 (FastGA.c:4190) */
int fga_reference_slots(const int64_t *counts, const int64_t *clen, int nctg, int nthreads, int swide, int *slot)
{ int *split, *beg, *end;
  int64_t *off, *panel, npost = 0, cum, t;
  int nparts, p, r, x, u, i;
  if (counts == NULL || clen == NULL || slot == NULL || nctg < 1 || nthreads < 1 || swide < 1)
    { fga_set_error("fga_reference_slots: bad argument");
      return 1;
    }
  split = malloc(sizeof(int)*(nthreads+2 + (size_t) nctg));
  beg = malloc(sizeof(int)*2*nthreads);
  off = malloc(sizeof(int64_t)*nthreads);
  panel = malloc(sizeof(int64_t)*nctg);
  if (split == NULL || beg == NULL || off == NULL || panel == NULL)
    { free(split); free(beg); free(off); free(panel);
      fga_set_error("out of memory");
      return 1;
    }
  end = beg + nthreads;
  /* the A-contig parts: consecutive runs of at least NTHREADS contigs and seqtot/NTHREADS bases (FastGA.c:5057-5086) */
  for (x = 0; x < nctg; x++) npost += clen[x];
  split[0] = 0;
  p = 0;
  r = nthreads;
  t = npost / nthreads;
  cum = clen[0];
  for (x = 1; x < nctg; x++)
    { if (cum >= t && x >= r)
        { p += 1;
          split[p] = x;
          t = (npost*(p+1)) / nthreads;
          r += nthreads;
        }
      cum += clen[x];
    }
  nparts = p+1;
  split[nparts] = nctg;
  for (x = 0; x < 2*nctg; x++) slot[x] = -1;
  for (u = 0; u < 2; u++)
    for (i = 0; i < nparts; i++)
      { int64_t nels = 0;
        int n;
        memset(panel,0,sizeof(int64_t)*nctg);
        for (x = split[i]; x < split[i+1]; x++)
          { panel[x] = counts[(size_t) u*nctg + x] * swide;
            nels += counts[(size_t) u*nctg + x];
          }
        n = fga_rmsd_ranges(panel,nctg,nels*swide,nthreads,beg,end,off);
        for (p = 0; p < n; p++)
          for (x = beg[p]; x < end[p]; x++)
            if (panel[x] > 0)
              slot[(size_t) u*nctg + x] = p;
      }
  free(split); free(beg); free(off); free(panel);
  return 0;
}

/* a set in the filter's order (aread, abpos, bread, comp, survival) -> the reference's order for the given slots:
   inside every run of equal (aread, abpos) the records of the strand with the smaller slot first (stable).
   invp[aread] = length-sorted index of original contig aread.  The trace bytes are laid out in record order again. */
int fga_alns_reference_order(fga_alns *A, const int *slot, const int *invp, int nctg)
{ int64_t i, j, k, tbcap = 0;
  fga_aln *tmp = NULL;
  uint8_t *tb = NULL;
  int64_t tmpcap = 0;
  int relay = 0;
  if (A == NULL || slot == NULL || invp == NULL)
    { fga_set_error("fga_alns_reference_order: null argument");
      return 1;
    }
  for (i = 0; i < A->naln; i = j)
    { const fga_aln *a = A->alns + i;
      int sn, sc, first, have0 = 0, have1 = 0;
      int64_t at;
      for (j = i; j < A->naln && A->alns[j].aread == a->aread && A->alns[j].abpos == a->abpos; j++)
        { if (A->alns[j].flags & 1) have1 = 1; else have0 = 1; }
      if (!(have0 && have1) || a->aread < 0 || a->aread >= nctg)
        continue;
      sn = slot[invp[a->aread]]; sc = slot[(size_t) nctg + invp[a->aread]];
      if (sn == sc)
        continue;
      first = sn < sc ? 0 : 1;                     /* the strand that goes first */
      /* is the run in that order already? */
      { int seen_other = 0, bad = 0;
        for (k = i; k < j; k++)
          { const int c = (int) (A->alns[k].flags & 1);
            if (c != first) seen_other = 1;
            else if (seen_other) { bad = 1; break; }
          }
        if (!bad) continue;
      }
      if (j - i > tmpcap)
        { fga_aln *nt;
          tmpcap = 2*(j-i) + 16;
          nt = realloc(tmp,sizeof(fga_aln)*tmpcap);
          if (nt == NULL) goto oom;
          tmp = nt;
        }
      at = 0;
      for (k = i; k < j; k++) if ((int) (A->alns[k].flags & 1) == first) tmp[at++] = A->alns[k];
      for (k = i; k < j; k++) if ((int) (A->alns[k].flags & 1) != first) tmp[at++] = A->alns[k];
      /* the run's trace bytes: when they lie in record order (what the filter and fga_alns_merge_filtered produce) they
         are permuted inside their own byte range; any other layout is re-laid as a whole at the end */
      { int64_t t0 = A->alns[i].toff, t1 = t0, tat;
        int packed = 1;
        for (k = i; k < j; k++)
          { if (A->alns[k].toff != t1) { packed = 0; break; }
            t1 += A->alns[k].tlen;
          }
        if (packed && !relay)
          { if (t1-t0 > tbcap)
              { uint8_t *nb;
                tbcap = 2*(t1-t0) + 4096;
                nb = realloc(tb,tbcap);
                if (nb == NULL) goto oom;
                tb = nb;
              }
            tat = 0;
            for (k = 0; k < j-i; k++)
              { memcpy(tb + tat,A->tbytes + tmp[k].toff,(size_t) tmp[k].tlen);
                tmp[k].toff = t0 + tat;
                tat += tmp[k].tlen;
              }
            memcpy(A->tbytes + t0,tb,(size_t) (t1-t0));
          }
        else
          relay = 1;
      }
      memcpy(A->alns + i,tmp,sizeof(fga_aln)*(j-i));
    }
  if (relay)
    { /* trace bytes in record order again (fga_alns_merge_filtered and the writers take runs of them) */
      int64_t tat = 0;
      uint8_t *all = malloc(A->ntrace + 16);
      if (all == NULL) goto oom;
      for (i = 0; i < A->naln; i++)
        { fga_aln *a = A->alns + i;
          memcpy(all + tat,A->tbytes + a->toff,(size_t) a->tlen);
          a->toff = tat;
          tat += a->tlen;
        }
      free(A->tbytes);
      A->tbytes = all;
      A->ntrace = tat;
    }
  for (i = 0; i < A->naln; i++)
    A->alns[i].seq = (int32_t) i;
  free(tmp); free(tb);
  return 0;
oom:
  free(tmp); free(tb);
  fga_set_error("out of memory ordering alignment records");
  return 1;
}
