/* fga_pipeline.c -- the FastGA hot path end to end on one GPU: the span of the reference's "Total Resources"
 * line (FastGA.c:4828-4829, 5263-5264): prebuilt GDB + GIX in, .1aln out.
 *
 *   phase 1  adaptive seed merge          adaptamer_merge / self_adaptamer_merge (FastGA.c:2281, 2496)   GPU
 *   phase 2  seed -> diagonal record sort reimport_thread + rmsd_sort (FastGA.c:2641, RSDsort.c:292)     GPU
 *            chain detection              align_contigs scan (FastGA.c:3016-3176)                        host threads
 *            wave extension               Local_Alignment loop (FastGA.c:3227-3341, align.c:1423)        GPU
 *            redundancy filter            FastGA.c:3405-3694                                             host
 *   phase 3  order + emit                 la_sort / la_merge (FastGA.c:3800-4133)                        host
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
  };

void fga_session_close(fga_session *Z)
{ if (Z == NULL) return;
  if (Z->dg2 != Z->dg1) fga_dgenome_free(Z->dg2);
  fga_dgenome_free(Z->dg1);
  fga_dgix_free(Z->d2); fga_dgix_free(Z->d1);
  fga_dev_close(Z->dev);
  fga_gix_close(Z->x2); fga_gix_close(Z->x1);
  fga_gdb_close(Z->g2); fga_gdb_close(Z->g1);
  free(Z);
}

static int gix_exists(const char *root)
{ char *p = NULL;
  size_t n = strlen(root);
  int ok;
  if (n > 4 && (strcmp(root+n-4,".gix") == 0 || strcmp(root+n-4,".gdb") == 0)) n -= 4;
  else if (n > 5 && strcmp(root+n-5,".1gdb") == 0) n -= 5;
  if (asprintf(&p,"%.*s.gix",(int) n,root) < 0) return 0;
  ok = access(p,R_OK) == 0;
  free(p);
  return ok;
}

/* load GDB + GIX of both genomes and make them resident in HBM */
int fga_session_open(const char *root1, const char *root2, int device, fga_session **out)
{ fga_session *Z = calloc(1,sizeof(fga_session));
  double t0;
  *out = NULL;
  if (Z == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  Z->self = (root2 == NULL);
  t0 = fga_wall();
  /* an index file is loaded when it is there; otherwise the index is built on the device from the GDB, straight
     into HBM (no .gix/.ktab files appear, like the reference without -k) */
  { int have1 = gix_exists(root1), have2 = Z->self ? 1 : gix_exists(root2);
    if (fga_gdb_open(root1,&Z->g1) || (have1 && fga_gix_open(root1,&Z->x1))) goto fail;
    if (!Z->self)
      { if (fga_gdb_open(root2,&Z->g2) || (have2 && fga_gix_open(root2,&Z->x2))) goto fail; }
    Z->load_s = fga_wall() - t0;
    if (fga_dev_open(device,&Z->dev)) goto fail;
    t0 = fga_wall();
    Z->devbuilt = !have1 || !have2;
    if (have1 ? fga_dgix_upload(Z->dev,Z->x1,&Z->d1) : fga_dgix_build(Z->dev,Z->g1,8,FGA_GIX_SOFT_MASK,&Z->d1,&Z->x1)) goto fail;
    if (!Z->self && (have2 ? fga_dgix_upload(Z->dev,Z->x2,&Z->d2) : fga_dgix_build(Z->dev,Z->g2,8,FGA_GIX_SOFT_MASK,&Z->d2,&Z->x2)))
      goto fail;
  }
  if (Z->x1->nctg < Z->g1->ncontig || (!Z->self && Z->x2->nctg < Z->g2->ncontig))
    { fga_set_error("genome index and genome database disagree on the number of contigs");
      goto fail;
    }
  if (fga_dgenome_upload(Z->dev,Z->g1,Z->x1->perm,Z->x1->nctg,1,&Z->dg1)) goto fail;
  if (Z->self)
    Z->dg2 = Z->dg1;
  else if (fga_dgenome_upload(Z->dev,Z->g2,Z->x2->perm,Z->x2->nctg,1,&Z->dg2)) goto fail;
  Z->upload_s = fga_wall() - t0;
  *out = Z;
  return 0;
fail:
  fga_session_close(Z);
  return 1;
}

fga_dev *fga_session_device(fga_session *Z) { return Z->dev; }
int64_t  fga_session_table_bytes(const fga_session *Z)
{ return Z->x1->nents*Z->x1->ebytes + (Z->self ? 0 : Z->x2->nents*Z->x2->ebytes); }
int      fga_session_seed_bytes(const fga_session *Z)
{ return 1 + Z->x1->postbytes + Z->x1->contbytes + (Z->self ? Z->x1->postbytes + Z->x1->contbytes
                                                            : Z->x2->postbytes + Z->x2->contbytes); }
int64_t  fga_session_bases(const fga_session *Z, int which)
{ return which == 0 ? Z->g1->seqtot : (Z->self ? Z->g1->seqtot : Z->g2->seqtot); }

/* one pass of the hot path over the resident inputs: phases 1-3 */
int fga_session_run(fga_session *Z, const fga_run_params *P, fga_run_stats *S)
{ fga_gdb *g1 = Z->g1, *g2 = Z->g2;
  fga_gix *x1 = Z->x1, *x2 = Z->x2;
  fga_dev *dev = Z->dev;
  fga_dgix *d1 = Z->d1, *d2 = Z->d2;
  fga_dgenome *dg1 = Z->dg1, *dg2 = Z->dg2;
  fga_dseeds *seeds = NULL;
  fga_dkeys *keys = NULL;
  fga_hits *hits = NULL;
  fga_alns *raw = NULL, *fin = NULL;
  void *hkeys = NULL;
  int64_t *alen = NULL;
  int16_t *table = NULL;
  const int self = Z->self;
  int status = 1, i;
  double t0, t1, tstart;
  fga_run_stats st;

  memset(&st,0,sizeof(st));
  st.load_s = Z->load_s; st.upload_s = Z->upload_s;
  tstart = fga_wall();
  /* ---- phase 1 ---- */
  t0 = fga_wall();
  { fga_merge_params mp;
    int rc;
    memset(&mp,0,sizeof(mp));
    mp.freq = P->freq; mp.soft_mask = P->soft_mask; mp.flip = 0;
    rc = fga_seed_merge(dev,d1,self ? NULL : d2,&mp,
                        (P->symmetric && !self) ? 2*(x1->nents + x2->nents) + (1<<20) : 0,&seeds);
    if (rc == 2)
      { int64_t need = fga_seeds_count(seeds) + 1024;
        fga_seeds_free(seeds); seeds = NULL;
        rc = fga_seed_merge(dev,d1,self ? NULL : d2,&mp,need,&seeds);
      }
    if (rc) goto done;
    if (P->symmetric && !self)
      { mp.flip = 1;
        rc = fga_seed_merge_append(dev,d2,d1,&mp,seeds);
        if (rc) goto done;
      }
    st.nseeds = fga_seeds_count(seeds);
    st.seed_len_sum = fga_seeds_plen_sum(seeds);
    if (self) { st.nseeds /= 2; st.seed_len_sum /= 2; }
  }
  st.merge_s = fga_wall() - t0;
  st.merge_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_MERGE);

  /* ---- phase 2 ---- */
  t0 = fga_wall();
  { fga_sort_params sp;
    sp.amxpos = g1->maxctg; sp.bmxpos = self ? g1->maxctg : g2->maxctg;
    sp.nctg_a = x1->nctg;   sp.nctg_b = self ? x1->nctg : x2->nctg;
    sp.anti_order_only = 1;          /* chains do not depend on the order inside an (anti-diagonal) tie */
    if (fga_seed_sort(dev,seeds,&sp,&keys)) goto done;
    fga_seeds_free(seeds); seeds = NULL;
    st.sort_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_SORT);
  }
  t1 = fga_wall();
  st.sort_s = t1 - t0;

  { int64_t n = fga_keys_count(keys);
    int wa, wb, wd, wt;
    fga_chain_params cp;
    fga_keys_layout(keys,&wa,&wb,&wd,&wt);
    (void) n;
    alen  = malloc(sizeof(int64_t)*x1->nctg);
    if (alen == NULL)
      { fga_set_error("out of memory");
        goto done;
      }
    for (i = 0; i < x1->nctg; i++)
      alen[i] = (x1->perm[i] < g1->ncontig) ? g1->contigs[x1->perm[i]].clen : FGA_KMER;
    cp.chain_break = P->chain_break; cp.chain_min = P->chain_min;
    cp.amxpos = g1->maxctg; cp.bmxpos = self ? g1->maxctg : g2->maxctg;
    cp.alen = alen; cp.nalen = x1->nctg;
    st.download_s = 0.;                              /* the sorted records never leave HBM any more */
    if (fga_chain_scan_device(dev,keys,&cp,&hits)) goto done;
    fga_keys_free(keys); keys = NULL;
    st.nhits = hits->nhits;
    st.nunits = hits->nunits;
    st.chain_s = fga_wall() - t1;
  }

  t1 = fga_wall();
  { fga_extend_params ep;
    int path_ave;
    memset(&ep,0,sizeof(ep));
    table = malloc(sizeof(int16_t)*2*32768);
    if (table == NULL)
      { fga_set_error("out of memory");
        goto done;
      }
    fga_align_spec(1.-P->align_rate,100,g1->freq,&path_ave,table,table+32768);
    ep.tspace = 100; ep.path_ave = path_ave; ep.table = table; ep.score = table+32768;
    ep.self = self; ep.aln_min = P->align_min - 50; ep.aln_rate = P->align_rate + .05;
    if (fga_extend(dev,dg1,dg2,hits,&ep,&raw)) goto done;
    st.nalns = raw->naln; st.ncalls = raw->ncalls; st.nwaves = raw->nwaves;
    st.extend_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_EXTEND);
  }
  st.extend_s = fga_wall() - t1;

  t1 = fga_wall();
  if (fga_filter_alignments_mt(raw,P->nthreads,&fin)) goto done;
  st.nlive = fin->naln;
  for (i = 0; i < fin->naln; i++)
    st.cover += fin->alns[i].aepos - fin->alns[i].abpos;
  st.filter_s = fga_wall() - t1;

  /* ---- phase 3 ---- */
  t1 = fga_wall();
  if (P->out_path != NULL)
    { char *n1 = NULL, *n2 = NULL;
      const char *cmd = P->command_line ? P->command_line : "FastGA";
      int rc;
      if (asprintf(&n1,"%s",g1->path) < 0) n1 = NULL;
      if (!self && asprintf(&n2,"%s",g2->path) < 0) n2 = NULL;
      /* the binary ONEcode container, like the reference; FGA_ALN_ASCII=1 selects the text form */
      if (getenv("FGA_ALN_ASCII") != NULL && atoi(getenv("FGA_ALN_ASCII")) != 0)
        rc = fga_write_1aln(P->out_path,g1,self ? NULL : g2,fin,100,n1 ? n1 : "genome1",n2,cmd);
      else
        rc = fga_write_1aln_binary(P->out_path,g1,self ? NULL : g2,fin,100,n1 ? n1 : "genome1",n2,cmd);
      free(n1); free(n2);
      if (rc) goto done;
    }
  st.write_s = fga_wall() - t1;
  st.phase23_s = fga_wall() - tstart;

  /* ---- PAF (what the reference leaves to a second process, ALNtoPAF): outside the .1aln clock ---- */
  if (P->paf_path != NULL)
    { fga_traces *tr = NULL;
      const int psl = (P->paf_flags & FGA_OUT_PSL) != 0;
      const int bases = psl || (P->paf_flags & (FGA_PAF_CIGAR_M|FGA_PAF_CIGAR_X|FGA_PAF_CS_SHORT|FGA_PAF_CS_LONG)) != 0;
      int rc = 0;
      t1 = fga_wall();
      if (bases)
        { rc = fga_trace_pts(dev,dg1,dg2,fin,100,0,&tr);
          st.trace_kernel_ms = fga_dev_stage_ms(dev,FGA_STAGE_TRACE);
        }
      st.trace_s = fga_wall() - t1;
      t1 = fga_wall();
      if (rc == 0)
        rc = psl ? fga_write_psl(P->paf_path,g1,self ? NULL : g2,fin,tr,P->nthreads)
                 : fga_write_paf(P->paf_path,g1,self ? NULL : g2,fin,tr,P->paf_flags,P->nthreads);
      st.paf_s = fga_wall() - t1;
      fga_traces_free(tr);
      if (rc) goto done;
    }
  status = 0;

done:
  if (S != NULL) *S = st;
  free(hkeys); free(alen); free(table);
  fga_alns_free(raw); fga_alns_free(fin);
  fga_hits_free(hits);
  fga_keys_free(keys);
  fga_seeds_free(seeds);
  return status;
}

int fga_run(const char *root1, const char *root2, const fga_run_params *P, fga_run_stats *S)
{ fga_session *Z;
  int rc;
  if (fga_session_open(root1,root2,P->device,&Z))
    return 1;
  rc = fga_session_run(Z,P,S);
  fga_session_close(Z);
  return rc;
}
