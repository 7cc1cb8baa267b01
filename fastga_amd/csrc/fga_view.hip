// fga_view.hip -- the genome index as the seed merge reads it: one array per field.
//
// On disk (and in fga_gix) a table entry is 9 + PostBytes + ContBytes bytes: [7 suffix | mask | lcp | post LE | contig|sign LE]
// (reference GIXmake.c:1100-1152) -- odd-width records that a wavefront can only take apart with
// unaligned dword reads, byte shuffles and byte swaps.  The kernel's view of a table in HBM is therefore five arrays over
// the entries in table order, made ONCE when the table reaches the device (fga_dgix_upload, fga_dgix_build):
//
//   K[i]  u64   (12-mer prefix & 0xff) << 56 | 56-bit suffix   -- keys compare across the panels of a tile: a tile never
//                                                                  crosses a multiple of 256 prefixes, so the order of K
//                                                                  is the order of the 40-mers
//   L[i]  u8    lcp byte; the first entry of a panel is clamped to <= 11 (its true lcp with the panel before is < 12;
//               the reference's GIXmake has been seen to leave a stale 12 there)
//   M[i]  u8    soft-mask byte (read by -M runs only)
//   P[i]  u32   position inside the contig
//   C[i]  u8 / u16 / u32   contig | sign, as stored (sign = top bit of the last byte)
//   idx[p] u32  inclusive cumulative entry count per 12-mer prefix, low 32 bits; car[k] = the first prefix at which the count
//               reaches (k+1) * 2^32 (tables of up to 5 * 2^32 entries: the reference indexes with int64, libfastk.c:785-907)
//
// = 14 + ContBytes bytes per entry of which a pair comparison reads K, P, C of table 1 and K, L, P, C of table 2: the
// on-disk width or one byte more.  The FORWARD view (fga_view_forward) holds the forward-strand entries of a table only,
// with their own prefix index: in a pair comparison the entries of table 1 on the complement strand produce nothing
// (FastGA.c:921-928: they are skipped after having been read), so the kernel does not read them at all.
#include "fga_device.hpp"
#include <atomic>

#define VW_T 256

__device__ __forceinline__ uint32_t vw_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{ return __builtin_amdgcn_alignbyte(hi,lo,sh); }

// one workgroup per 256 consecutive prefixes: every entry of their panels is taken apart
__global__ __launch_bounds__(VW_T)
void view_repack_kernel(const uint8_t *tab, const int64_t *idx64, int E, int post, int cont, fga_view V)
{ __shared__ int64_t e[257];
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t) blockIdx.x*256;
  e[tid+1] = idx64[p0 + tid];
  if (tid == 0)
    e[0] = p0 > 0 ? idx64[p0-1] : 0;
  __syncthreads();
  const int64_t lo = e[0], hi = e[256];
  const uint32_t pm = post >= 4 ? 0xffffffffu : ((1u << (8*post)) - 1);
  for (int64_t i = lo + tid; i < hi; i += VW_T)
    { int k = 0;                                       // panel of entry i: smallest k with e[k+1] > i
      #pragma unroll
      for (int s = 128; s >= 1; s >>= 1)
        if (e[k+s] <= i) k += s;
      const uint64_t addr = (uint64_t) (tab + i*E);
      const uint32_t *w = (const uint32_t *) (addr & ~(uint64_t) 3);
      const uint32_t sh = (uint32_t) (addr & 3);
      const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4], d5 = w[5];
      const uint32_t b0 = vw_alignbyte(d1,d0,sh), b1 = vw_alignbyte(d2,d1,sh), b2 = vw_alignbyte(d3,d2,sh),
                     b3 = vw_alignbyte(d4,d3,sh), b4 = vw_alignbyte(d5,d4,sh);
      // bytes 0..6: suffix, first base in the high bits
      const uint64_t be = ((uint64_t) __builtin_bswap32(b0) << 32) | __builtin_bswap32(b1);      // bytes 0..7 big endian
      const uint64_t suf = be >> 8;
      V.K[i] = ((uint64_t) ((p0 + k) & 0xff) << 56) | suf;
      V.M[i] = (uint8_t) (be & 0xff);
      uint32_t lcp = b2 & 0xff;
      if (i == e[k] && lcp > 11) lcp = 11;
      V.L[i] = (uint8_t) lcp;
      const uint64_t pay = ((((uint64_t) b3 << 32) | b2) >> 8) | ((uint64_t) (b4 & 0xffu) << 56);   // bytes 9..16 (an entry of 17 bytes at most)
      V.P[i] = (uint32_t) pay & pm;
      const uint32_t c = (uint32_t) (pay >> (8*post)) & (cont >= 4 ? 0xffffffffu : ((1u << (8*cont)) - 1));
      if (V.cw == 1)      ((uint8_t  *) V.C)[i] = (uint8_t) c;
      else if (V.cw == 2) ((uint16_t *) V.C)[i] = (uint16_t) c;
      else                ((uint32_t *) V.C)[i] = c;
    }
  V.idx[p0 + tid] = (uint32_t) e[tid+1];
}

__device__ __forceinline__ uint32_t view_c(const fga_view &V, int64_t i)
{ if (V.cw == 1) return ((const uint8_t  *) V.C)[i];
  if (V.cw == 2) return ((const uint16_t *) V.C)[i];
  return ((const uint32_t *) V.C)[i];
}

// forward view, pass 1: forward-strand entries per block of 1024 and per 64-entry piece of it
__global__ __launch_bounds__(VW_T)
void view_fwd_count_kernel(fga_view V, uint32_t signbit, uint32_t *blkcnt, uint16_t *sub)
{ __shared__ uint32_t wtot[VW_T/64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t) blockIdx.x*1024;
  uint32_t pc[4];
  #pragma unroll
  for (int r = 0; r < 4; r++)                      // piece r*4 + wave of the block: 64 consecutive entries, one per lane
    { const int64_t i = base + (int64_t) (r*4 + wave)*64 + lane;
      const bool f = i < V.n && !(view_c(V,i) & signbit);
      pc[r] = (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(f));
    }
  // exclusive offsets of the 16 pieces in piece order (piece q = r*4 + wave)
  if (lane == 0)
    { for (int r = 0; r < 4; r++) sub[(int64_t) blockIdx.x*16 + r*4 + wave] = (uint16_t) pc[r]; }
  __syncthreads();
  if (tid == 0)
    { uint32_t run = 0;
      for (int q = 0; q < 16; q++)
        { const uint16_t c = sub[(int64_t) blockIdx.x*16 + q];
          sub[(int64_t) blockIdx.x*16 + q] = (uint16_t) run;
          run += c;
        }
      blkcnt[blockIdx.x] = run;
    }
  (void) wtot;
}

// pass 2: the forward entries of every 64-entry piece go to blkoff[block] + sub[piece] + rank inside the piece
__global__ __launch_bounds__(VW_T)
void view_fwd_scatter_kernel(fga_view V, fga_view F, uint32_t signbit, const int64_t *blkoff, const uint16_t *sub)
{ const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t) blockIdx.x*1024;
  const int64_t boff = blkoff[blockIdx.x];
  #pragma unroll
  for (int r = 0; r < 4; r++)
    { const int q = r*4 + wave;
      const int64_t i = base + (int64_t) q*64 + lane;
      const uint32_t c = i < V.n ? view_c(V,i) : signbit;
      const bool f = !(c & signbit);
      const uint64_t m = __builtin_amdgcn_ballot_w64(f);
      if (f)
        { const int64_t o = boff + sub[(int64_t) blockIdx.x*16 + q] + __popcll(m & ((1ull << lane) - 1));
          F.K[o] = V.K[i]; F.M[o] = V.M[i]; F.P[o] = V.P[i];
          if (F.cw == 1)      ((uint8_t  *) F.C)[o] = (uint8_t) c;
          else if (F.cw == 2) ((uint16_t *) F.C)[o] = (uint16_t) c;
          else                ((uint32_t *) F.C)[o] = c;
        }
    }
}

// forward entries among the entries of the prefixes <= p
__device__ __forceinline__ int64_t view_fwd_upto(const fga_view &V, const fga_car &car, int64_t fn, uint32_t signbit,
                                                 const int64_t *blkoff, const uint16_t *sub, int64_t p)
{ const int64_t e = fga_idx_abs(V.idx,car,p);         // entries [0,e) belong to prefixes <= p
  if (e >= V.n)
    return fn;
  const int64_t piece = e >> 6;
  int64_t r = blkoff[piece >> 4] + sub[piece];
  for (int64_t i = piece << 6; i < e; i++)
    r += !(view_c(V,i) & signbit);
  return r;
}

// pass 3: the forward view's prefix index: forward entries before the end of every panel (low words; the carries below)
__global__ __launch_bounds__(VW_T)
void view_fwd_index_kernel(fga_view V, fga_car car, fga_view F, uint32_t signbit, const int64_t *blkoff, const uint16_t *sub)
{ const int64_t p = (int64_t) blockIdx.x*VW_T + threadIdx.x;
  if (p >= FGA_NPREFIX)
    return;
  F.idx[p] = (uint32_t) view_fwd_upto(V,car,F.n,signbit,blkoff,sub,p);
}

// car[k] of a cumulative count given as a function of the prefix: the first prefix at which it reaches (k+1) * 2^32
__global__ void view_fwd_carry_kernel(fga_view V, fga_car car, int64_t fn, uint32_t signbit, const int64_t *blkoff, const uint16_t *sub,
                                      uint32_t *out)
{ const int k = threadIdx.x;
  if (k >= 4) return;
  const int64_t want = (int64_t) (k+1) << 32;
  int64_t lo = 0, hi = FGA_NPREFIX;                  // smallest p with count(p) >= want, or 2^24
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      if (view_fwd_upto(V,car,fn,signbit,blkoff,sub,mid) >= want) hi = mid; else lo = mid + 1;
    }
  out[k] = (uint32_t) lo;
}

__global__ void view_carry_kernel(const int64_t *idx64, uint32_t *out)
{ const int k = threadIdx.x;
  if (k >= 4) return;
  const int64_t want = (int64_t) (k+1) << 32;
  int64_t lo = 0, hi = FGA_NPREFIX;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      if (idx64[mid] >= want) hi = mid; else lo = mid + 1;
    }
  out[k] = (uint32_t) lo;
}

int fga_view_set_carries(fga_dev *dev, const int64_t *idx64, fga_view *V)
{ uint32_t *d = NULL;
  for (int k = 0; k < 4; k++) V->car[k] = (uint32_t) FGA_NPREFIX;
  if (V->n < ((int64_t) 1 << 32))
    return 0;
  if (fga_dmalloc(&d,sizeof(uint32_t)*4) != hipSuccess)
    { fga_set_error("device allocation failed"); return 1; }
  hipLaunchKernelGGL(view_carry_kernel,dim3(1),dim3(64),0,dev->stream,idx64,d);
  hipError_t e = hipMemcpyAsync(V->car,d,sizeof(uint32_t)*4,hipMemcpyDeviceToHost,dev->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
  fga_pool_free(d);
  if (e != hipSuccess)
    { fga_set_error("table view: carries: %s",hipGetErrorString(e)); return 1; }
  return 0;
}

static void view_free(fga_view *V)
{ fga_pool_free(V->K); fga_pool_free(V->L); fga_pool_free(V->M); fga_pool_free(V->P); fga_pool_free(V->C); fga_pool_free(V->idx);
  memset(V,0,sizeof(*V));
}

int fga_view_alloc(fga_view *V, int64_t n, int cont, int want_l)
{ static std::atomic<uint64_t> next_gen{1};
  memset(V,0,sizeof(*V));
  V->n = n;
  V->gen = next_gen.fetch_add(1);
  for (int k = 0; k < 4; k++) V->car[k] = (uint32_t) FGA_NPREFIX;
  V->cw = cont <= 1 ? 1 : (cont == 2 ? 2 : 4);
  const size_t m = (size_t) n + 256;                 // windows are loaded in 16-byte pieces from 4-entry aligned starts
  hipError_t e;
  if ((e = fga_dmalloc(&V->K,8*m)) != hipSuccess || (want_l && (e = fga_dmalloc(&V->L,m)) != hipSuccess) ||
      (e = fga_dmalloc(&V->M,m)) != hipSuccess || (e = fga_dmalloc(&V->P,4*m)) != hipSuccess ||
      (e = fga_dmalloc(&V->C,(size_t) V->cw*m)) != hipSuccess ||
      (e = fga_dmalloc(&V->idx,sizeof(uint32_t)*(size_t) FGA_NPREFIX)) != hipSuccess)
    { fga_set_error("device allocation of a table view (%lld entries) failed: %s",(long long) n,hipGetErrorString(e));
      view_free(V);
      return 1;
    }
  // the slack behind the last entry is read (never used): keep it defined
  // (on the calling context's stream: the kernels that fill the view follow on it)
  fga_memset_here(V->K + n,0xff,8*256);
  if (V->L) fga_memset_here(V->L + n,0,256);
  fga_memset_here(V->M + n,0,256); fga_memset_here(V->P + n,0,4*256); fga_memset_here((uint8_t *) V->C + (size_t) V->cw*n,0,(size_t) V->cw*256);
  return 0;
}

// D->table (the on-disk bytes) + D->index -> D->view; the on-disk bytes leave the device afterwards unless keep_table
int fga_dgix_make_view(fga_dev *dev, fga_dgix *D, int keep_table)
{ if (D->nents >= ((int64_t) 5 << 32) - 1024)
    { fga_set_error("genome index of %lld entries: tables of 5 x 2^32 entries and more are not supported",(long long) D->nents);
      return 1;
    }
  if (D->postbytes > 4)
    { fga_set_error("genome index with %d position bytes: contigs beyond 4 Gbp are not supported",D->postbytes);
      return 1;
    }
  if (D->postbytes + D->contbytes > 8)
    { fga_set_error("genome index with %d position and %d contig bytes: payloads beyond 8 bytes are not supported",
                    D->postbytes,D->contbytes);
      return 1;
    }
  double t0 = fga_wall();
  if (fga_view_alloc(&D->view,D->nents,D->contbytes,1))
    return 1;
  fga_note("view: allocation",t0); t0 = fga_wall();
  hipLaunchKernelGGL(view_repack_kernel,dim3(FGA_NPREFIX/256),dim3(VW_T),0,dev->stream,
                     D->table,D->index,D->ebytes,D->postbytes,D->contbytes,D->view);
  hipError_t e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("table view: kernel failed: %s",hipGetErrorString(e));
      view_free(&D->view);
      return 1;
    }
  fga_note("view: repack kernel",t0); t0 = fga_wall();
  if (fga_view_set_carries(dev,D->index,&D->view))
    { view_free(&D->view);
      return 1;
    }
  if (!keep_table)
    { fga_pool_free(D->table);
      D->table = NULL;
      fga_note("view: on-disk bytes freed",t0);
    }
  return 0;
}

// the forward-strand entries of D->view as a table of their own (made on first use as table 1 of a pair comparison)
int fga_dgix_make_forward(fga_dev *dev, fga_dgix *D)
{ if (D->fview.K != NULL)
    return 0;
  const double tf0 = fga_wall();
  const fga_view &V = D->view;
  const uint32_t signbit = 0x80u << (8*(D->contbytes-1));
  const int64_t nblk = (V.n + 1023) / 1024;
  uint32_t *dcnt = NULL;
  uint16_t *dsub = NULL;
  int64_t *doff = NULL;
  int rc = 1;
  hipError_t e;
  std::vector<uint32_t> cnt((size_t) nblk + 1);
  std::vector<int64_t> off((size_t) nblk + 1);
  int64_t nf = 0;
  fga_view F;
  memset(&F,0,sizeof(F));
  if ((e = fga_dmalloc(&dcnt,sizeof(uint32_t)*(size_t) (nblk+1))) != hipSuccess ||
      (e = fga_dmalloc(&dsub,sizeof(uint16_t)*16*(size_t) (nblk+1))) != hipSuccess ||
      (e = fga_dmalloc(&doff,sizeof(int64_t)*(size_t) (nblk+1))) != hipSuccess)
    { fga_set_error("forward view: device allocation failed: %s",hipGetErrorString(e));
      goto done;
    }
  if (nblk > 0)
    hipLaunchKernelGGL(view_fwd_count_kernel,dim3((unsigned) nblk),dim3(VW_T),0,dev->stream,V,signbit,dcnt,dsub);
  if ((e = hipMemcpyAsync(cnt.data(),dcnt,sizeof(uint32_t)*(size_t) nblk,hipMemcpyDeviceToHost,dev->stream)) != hipSuccess ||
      (e = hipStreamSynchronize(dev->stream)) != hipSuccess)
    { fga_set_error("forward view: count failed: %s",hipGetErrorString(e));
      goto done;
    }
  for (int64_t b = 0; b < nblk; b++)
    { off[(size_t) b] = nf; nf += cnt[(size_t) b]; }
  off[(size_t) nblk] = nf;
  if (fga_view_alloc(&F,nf,D->contbytes,0))
    goto done;
  if ((e = hipMemcpyAsync(doff,off.data(),sizeof(int64_t)*(size_t) (nblk+1),hipMemcpyHostToDevice,dev->stream)) != hipSuccess)
    { fga_set_error("forward view: upload failed: %s",hipGetErrorString(e));
      goto done;
    }
  if (nblk > 0)
    hipLaunchKernelGGL(view_fwd_scatter_kernel,dim3((unsigned) nblk),dim3(VW_T),0,dev->stream,V,F,signbit,doff,dsub);
  hipLaunchKernelGGL(view_fwd_index_kernel,dim3(FGA_NPREFIX/VW_T),dim3(VW_T),0,dev->stream,V,fga_view_car(V),F,signbit,doff,dsub);
  if (nf >= ((int64_t) 1 << 32))            // the forward view's own carries
    { uint32_t *dc = NULL;
      if ((e = fga_dmalloc(&dc,sizeof(uint32_t)*4)) != hipSuccess)
        { fga_set_error("forward view: device allocation failed: %s",hipGetErrorString(e));
          goto done;
        }
      hipLaunchKernelGGL(view_fwd_carry_kernel,dim3(1),dim3(64),0,dev->stream,V,fga_view_car(V),nf,signbit,doff,dsub,dc);
      e = hipMemcpyAsync(F.car,dc,sizeof(uint32_t)*4,hipMemcpyDeviceToHost,dev->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
      fga_pool_free(dc);
      if (e != hipSuccess)
        { fga_set_error("forward view: carries: %s",hipGetErrorString(e));
          goto done;
        }
    }
  e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("forward view: kernel failed: %s",hipGetErrorString(e));
      goto done;
    }
  D->fview = F;
  memset(&F,0,sizeof(F));
  fga_note("forward view",tf0);
  rc = 0;
done:
  fga_pool_free(dcnt); fga_pool_free(dsub); fga_pool_free(doff);
  if (rc) view_free(&F);
  return rc;
}

void fga_dgix_free_views(fga_dgix *D)
{ view_free(&D->view);
  view_free(&D->fview);
  fga_pool_free(D->cutc.cuts);          // the cached range cuts name the views' prefix indices
  memset(&D->cutc,0,sizeof(D->cutc));
}
