// fga_shard.hip -- routing seeds to A-contig parts on the device (one comparison over several GPUs, or over several
// passes of one GPU).
//
// Replaces the reference's seed file matrix: every merge thread appends a seed to the file of (its own slot,
// Select[A contig]) and counts it in buck[] (FastGA.c:933-984, 5097-5134), and the search phase re-reads the files of
// one part (the "transpose", FastGA.c:5160-5184, 4160-4187).  Here a rank's seeds stay in HBM:
//   fga_seeds_contig_histogram  = buck[]: seeds per A contig (LDS-privatised counters, one flush per workgroup),
//   fga_seeds_split_to          = the transpose: seeds regrouped by Select[A contig] into a caller-provided device
//                                 buffer (the send buffer of the RCCL all-to-all-v, or the per-part staging of a
//                                 single-GPU multi-pass run): a counting pass + a scatter pass, 16-byte records, both
//                                 HBM-bound (2 reads + 1 write of S x 16 B),
//   fga_seeds_import            = the receiving side: pieces from several ranks become one seed buffer.
#include "fga_device.hpp"

#define SH_T     256
#define SH_TILE  8192                // seeds per workgroup of the split passes
#define SH_LDSH  8192                // LDS-privatised contig counters
#define SH_MAXP  64                  // parts

__global__ __launch_bounds__(SH_T)
void seed_contig_hist_kernel(const fga_seed *seeds, int64_t n, int nctg, int strands, unsigned long long *counts, const uint16_t *valid)
{ __shared__ unsigned int h[SH_LDSH];
  const int nh = strands ? 2*nctg : nctg;       // strands: the C-stream seeds are counted in a second row of nctg counters
  const bool lds = nh <= SH_LDSH;
  if (lds)
    { for (int c = threadIdx.x; c < nh; c += SH_T) h[c] = 0;
      __syncthreads();
    }
  // a workgroup takes consecutive 16 KB stretches; its LDS counters cannot overflow (< 2^32 seeds per workgroup)
  for (int64_t i = (int64_t) blockIdx.x*SH_T + threadIdx.x; i < n; i += (int64_t) gridDim.x*SH_T)
    { if (valid != NULL && (int) (i & 1023) >= (int) valid[i >> 10])       // a hole of the merge kernel's block allocation
        continue;
      const fga_seed s = seeds[i];
      uint32_t c = s.actg >> 8;
      if (c < (uint32_t) nctg)
        { if (strands && (s.bctg >> 31)) c += (uint32_t) nctg;
          if (lds) atomicAdd(h+c,1u);
          else     atomicAdd(counts+c,1ull);
        }
    }
  if (lds)
    { __syncthreads();
      for (int c = threadIdx.x; c < nh; c += SH_T)
        if (h[c] != 0) atomicAdd(counts+c,(unsigned long long) h[c]);
    }
}

// pass 1: seeds of tile b per part -> tilecnt[b*nparts + p]
__global__ __launch_bounds__(SH_T)
void seed_part_count_kernel(const fga_seed *seeds, int64_t n, const int *select, int nctg, int nparts,
                            unsigned int *tilecnt, const uint16_t *valid)
{ __shared__ unsigned int c[SH_MAXP];
  if (threadIdx.x < SH_MAXP) c[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t) blockIdx.x*SH_TILE;
  for (int k = threadIdx.x; k < SH_TILE; k += SH_T)
    { const int64_t i = base + k;
      if (i < n && (valid == NULL || (int) (i & 1023) < (int) valid[i >> 10]))
        { const uint32_t a = seeds[i].actg >> 8;
          const int p = a < (uint32_t) nctg ? select[a] : 0;
          atomicAdd(c+p,1u);
        }
    }
  __syncthreads();
  if ((int) threadIdx.x < nparts)
    tilecnt[(int64_t) blockIdx.x*nparts + threadIdx.x] = c[threadIdx.x];
}

// pass 2: tileoff[b*nparts + p] = first slot of (part p, tile b) in dst; order inside is arbitrary (the sort follows)
__global__ __launch_bounds__(SH_T)
void seed_part_scatter_kernel(const fga_seed *seeds, int64_t n, const int *select, int nctg, int nparts,
                              const int64_t *tileoff, fga_seed *dst, const uint16_t *valid)
{ __shared__ unsigned int cur[SH_MAXP];
  __shared__ int64_t off[SH_MAXP];
  if (threadIdx.x < SH_MAXP)
    { cur[threadIdx.x] = 0;
      off[threadIdx.x] = (int) threadIdx.x < nparts ? tileoff[(int64_t) blockIdx.x*nparts + threadIdx.x] : 0;
    }
  __syncthreads();
  const int64_t base = (int64_t) blockIdx.x*SH_TILE;
  for (int k = threadIdx.x; k < SH_TILE; k += SH_T)
    { const int64_t i = base + k;
      if (i < n && (valid == NULL || (int) (i & 1023) < (int) valid[i >> 10]))
        { const fga_seed s = seeds[i];
          const uint32_t a = s.actg >> 8;
          const int p = a < (uint32_t) nctg ? select[a] : 0;
          const unsigned int r = atomicAdd(cur+p,1u);
          dst[off[p] + r] = s;
        }
    }
}

extern "C" const void *fga_seeds_device_ptr(const fga_dseeds *S) { return S == NULL ? NULL : S->seeds; }

static int seeds_histogram(fga_dev *dev, const fga_dseeds *S, int nctg, int strands, int64_t *counts);

extern "C" int fga_seeds_contig_histogram(fga_dev *dev, const fga_dseeds *S, int nctg, int64_t *counts)
{ return seeds_histogram(dev,S,nctg,0,counts); }

// the same per strand: counts[u*nctg + j] = seeds of stream u (0: N, 1: C) whose A contig is j -- the reference's buck[]
// of its N_Units / C_Units (FastGA.c:933-984), what rmsd_sort cuts the search threads' ranges from (fga_order.c)
extern "C" int fga_seeds_strand_histogram(fga_dev *dev, const fga_dseeds *S, int nctg, int64_t *counts)
{ return seeds_histogram(dev,S,nctg,1,counts); }

static int seeds_histogram(fga_dev *dev, const fga_dseeds *S, int nctg_, int strands, int64_t *counts)
{ if (dev == NULL || S == NULL || counts == NULL || nctg_ <= 0)
    { fga_set_error("fga_seeds_contig_histogram: bad argument");
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  const int64_t n = fga_seeds_extent(S);
  const int nctg = strands ? 2*nctg_ : nctg_;          // counters
  unsigned long long *d = (unsigned long long *) fga_dev_acquire(dev,SLOT_MISC,sizeof(unsigned long long)*(size_t) nctg);
  if (d == NULL)
    { fga_set_error("fga_seeds_contig_histogram: device allocation failed");
      return 1;
    }
  hipMemsetAsync(d,0,sizeof(unsigned long long)*(size_t) nctg,dev->stream);
  if (n > 0)
    { int64_t wg = (n + SH_TILE - 1) / SH_TILE;
      if (wg > (int64_t) dev->ncu*8) wg = (int64_t) dev->ncu*8;
      hipLaunchKernelGGL(seed_contig_hist_kernel,dim3((unsigned) wg),dim3(SH_T),0,dev->stream,S->seeds,n,nctg_,strands,d,S->valid);
    }
  hipError_t e = hipMemcpyAsync(counts,d,sizeof(int64_t)*(size_t) nctg,hipMemcpyDeviceToHost,dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  fga_dev_release(dev,SLOT_MISC,d);
  if (e != hipSuccess)
    { fga_set_error("fga_seeds_contig_histogram: %s",hipGetErrorString(e));
      return 1;
    }
  return 0;
}

extern "C" int fga_seeds_split_to(fga_dev *dev, const fga_dseeds *S, const int *select, int nctg, int nparts,
                                  void *dst_device, int64_t *part_off)
{ if (dev == NULL || S == NULL || select == NULL || part_off == NULL || nctg <= 0 || nparts < 1 || nparts > SH_MAXP)
    { fga_set_error("fga_seeds_split_to: bad argument (1 <= nparts <= %d)",SH_MAXP);
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  const int64_t n = fga_seeds_extent(S);
  for (int p = 0; p <= nparts; p++) part_off[p] = 0;
  if (n == 0)
    return 0;
  if (dst_device == NULL)
    { fga_set_error("fga_seeds_split_to: no destination buffer");
      return 1;
    }
  const int64_t nt = (n + SH_TILE - 1) / SH_TILE;
  const size_t cbytes = sizeof(unsigned int)*(size_t) nt*nparts, obytes = sizeof(int64_t)*(size_t) nt*nparts;
  const size_t sbytes = sizeof(int)*(size_t) nctg;
  uint8_t *w = (uint8_t *) fga_dev_acquire(dev,SLOT_MISC,cbytes + obytes + sbytes + 64);
  if (w == NULL)
    { fga_set_error("fga_seeds_split_to: device allocation failed");
      return 1;
    }
  int64_t *d_off = (int64_t *) w;                                   // 8-byte aligned first
  unsigned int *d_cnt = (unsigned int *) (w + obytes);
  int *d_sel = (int *) (w + obytes + cbytes);
  std::vector<unsigned int> cnt((size_t) nt*nparts);
  std::vector<int64_t> off((size_t) nt*nparts);
  hipError_t e = hipMemcpyAsync(d_sel,select,sbytes,hipMemcpyHostToDevice,dev->stream);
  hipLaunchKernelGGL(seed_part_count_kernel,dim3((unsigned) nt),dim3(SH_T),0,dev->stream,S->seeds,n,d_sel,nctg,nparts,d_cnt,S->valid);
  if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(),d_cnt,cbytes,hipMemcpyDeviceToHost,dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess)
    { // part-major exclusive scan over (part, tile): a few 100 K integers
      int64_t run = 0;
      for (int p = 0; p < nparts; p++)
        { part_off[p] = run;
          for (int64_t b = 0; b < nt; b++)
            { off[(size_t) (b*nparts + p)] = run;
              run += cnt[(size_t) (b*nparts + p)];
            }
        }
      part_off[nparts] = run;
      e = hipMemcpyAsync(d_off,off.data(),obytes,hipMemcpyHostToDevice,dev->stream);
      hipLaunchKernelGGL(seed_part_scatter_kernel,dim3((unsigned) nt),dim3(SH_T),0,dev->stream,S->seeds,n,d_sel,nctg,nparts,
                         d_off,(fga_seed *) dst_device,S->valid);
      if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
      if (e == hipSuccess) e = hipGetLastError();
    }
  fga_dev_release(dev,SLOT_MISC,w);
  if (e != hipSuccess)
    { fga_set_error("fga_seeds_split_to: %s",hipGetErrorString(e));
      return 1;
    }
  return 0;
}

static int seeds_import(fga_dev *dev, const void *const *src_device, const int *src_ids, const int64_t *counts, int npieces,
                        fga_dseeds **out);

extern "C" int fga_seeds_import(fga_dev *dev, const void *const *src_device, const int64_t *counts, int npieces,
                                fga_dseeds **out)
{ return seeds_import(dev,src_device,NULL,counts,npieces,out); }

// a seed buffer that IS a dense stretch of device memory the caller keeps (one part's piece of a fga_seeds_split_to buffer on
// the same device): nothing is copied, fga_seeds_free leaves the stretch alone -- the caller releases it when the consumers
// (fga_session_align reads its seeds once, in the sort's first pass) are through
extern "C" int fga_seeds_view(fga_dev *dev, const void *src_device, int64_t count, fga_dseeds **out)
{ *out = NULL;
  if (dev == NULL || count < 0 || (count > 0 && src_device == NULL))
    { fga_set_error("fga_seeds_view: bad argument");
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  fga_dseeds *S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
  if (S == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  S->dev = dev; S->capacity = S->phys_capacity = count; S->count = S->phys_count = count; S->tseed = 0;
  S->seeds = (fga_seed *) src_device;
  S->slot = SLOT_BORROWED;
  hipError_t e = fga_dmalloc(&S->dcount,4*sizeof(unsigned long long));
  if (e == hipSuccess)
    { unsigned long long hc[4] = { (unsigned long long) count, 0ull, 0ull, 0ull };
      e = hipMemcpyAsync(S->dcount,hc,sizeof(hc),hipMemcpyHostToDevice,dev->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    }
  if (e != hipSuccess)
    { fga_set_error("fga_seeds_view: %s",hipGetErrorString(e));
      fga_seeds_free(S);
      return 1;
    }
  *out = S;
  return 0;
}

// the same for pieces that lie in the memory of OTHER devices of the node (src_device_id[k] = HIP device of piece k): the
// receiving side of the seed exchange of fga_run_multi, hipMemcpyPeerAsync over xGMI
extern "C" int fga_seeds_import_peer(fga_dev *dev, const void *const *src_device, const int *src_device_id,
                                     const int64_t *counts, int npieces, fga_dseeds **out)
{ if (npieces > 0 && src_device_id == NULL)
    { fga_set_error("fga_seeds_import_peer: bad argument");
      *out = NULL;
      return 1;
    }
  return seeds_import(dev,src_device,src_device_id,counts,npieces,out);
}

static int seeds_import(fga_dev *dev, const void *const *src_device, const int *src_ids, const int64_t *counts, int npieces,
                        fga_dseeds **out)
{ *out = NULL;
  if (dev == NULL || npieces < 0 || (npieces > 0 && (src_device == NULL || counts == NULL)))
    { fga_set_error("fga_seeds_import: bad argument");
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  int64_t total = 0;
  for (int k = 0; k < npieces; k++)
    { if (counts[k] < 0)
        { fga_set_error("fga_seeds_import: negative piece size");
          return 1;
        }
      total += counts[k];
    }
  fga_dseeds *S = (fga_dseeds *) calloc(1,sizeof(fga_dseeds));
  if (S == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  S->dev = dev; S->capacity = S->phys_capacity = total + 16; S->count = S->phys_count = total; S->tseed = 0;
  S->seeds = (fga_seed *) fga_dev_acquire(dev,SLOT_SEEDS,sizeof(fga_seed)*(size_t) S->phys_capacity);
  S->slot = SLOT_SEEDS;
  hipError_t e = fga_dmalloc(&S->dcount,4*sizeof(unsigned long long));
  if (S->seeds == NULL || e != hipSuccess)
    { fga_set_error("fga_seeds_import: device allocation failed");
      fga_dev_release(dev,SLOT_SEEDS,S->seeds); fga_pool_free(S->dcount); free(S);
      return 1;
    }
  { unsigned long long hc[4] = { (unsigned long long) total, 0ull, 0ull, 0ull };
    e = hipMemcpyAsync(S->dcount,hc,sizeof(hc),hipMemcpyHostToDevice,dev->stream);
  }
  int64_t at = 0;
  for (int k = 0; k < npieces && e == hipSuccess; k++)
    { if (counts[k] > 0)
        e = (src_ids != NULL && src_ids[k] != dev->device)
              ? hipMemcpyPeerAsync(S->seeds + at,dev->device,src_device[k],src_ids[k],sizeof(fga_seed)*(size_t) counts[k],dev->stream)
              : hipMemcpyAsync(S->seeds + at,src_device[k],sizeof(fga_seed)*(size_t) counts[k],hipMemcpyDeviceToDevice,dev->stream);
      at += counts[k];
    }
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  if (e != hipSuccess)
    { fga_set_error("fga_seeds_import: copy failed: %s",hipGetErrorString(e));
      fga_seeds_free(S);
      return 1;
    }
  *out = S;
  return 0;
}
