// tools/ubench/sort_bench.hip -- the radix sort of 128-bit records alone (fga_dev_radix_sort_u128 of the C-ABI), on keys
// made on the device: per-sort time, GB/s of algorithmic traffic (2 x n x 16 B per pass), sortedness and a checksum.
//   sort_bench <n keys> <nbits> <lowbit> <dist: uniform|seeds> <reps> [passes for the GB/s figure]
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/sort_bench.hip -o build/sort_bench -ldl
// The library is taken from $FGA_LIBRARY or fastga_amd/libfastga_amd.so (A/B variants: tools/build_variant.sh).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr,"%s: %s\n",#x,hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{ x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// dist 0: uniform bits below `top`; dist 1: seed-like -- [strand|A contig 5|B contig 5|diag bucket 22|anti 28|12], 70 % of the keys on
// a few diagonal buckets of homologous contig pairs (what a pair comparison of two 3 Gbp genomes produces)
__global__ void fill_kernel(uint4 *k, int64_t n, int top, int dist, uint64_t seed)
{ for (int64_t i = blockIdx.x*(int64_t) blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x*blockDim.x)
    { uint64_t a = mix(seed + 2*i), b = mix(seed + 2*i + 1);
      unsigned __int128 v;
      if (dist == 0)
        v = ((unsigned __int128) b << 64) | a;
      else
        { const uint64_t low = a & 0xfff, anti = (a >> 12) & ((1ull << 28) - 1);
          uint64_t ac = (a >> 40) & 31, bc = (a >> 45) & 31, db = b & ((1ull << 22) - 1), st = (b >> 22) & 1;
          if ((b >> 32) % 10 < 7)
            { bc = ac; st = 0; db = (1ull << 21) + ((b >> 40) & 3); }
          v = low | ((unsigned __int128) anti << 12) | ((unsigned __int128) db << 40) | ((unsigned __int128) bc << 62)
                  | ((unsigned __int128) ac << 67) | ((unsigned __int128) st << 72);
        }
      if (top < 128)
        v &= (((unsigned __int128) 1) << top) - 1;
      k[i] = make_uint4((uint32_t) v,(uint32_t) (v >> 32),(uint32_t) (v >> 64),(uint32_t) (v >> 96));
    }
}

__global__ void check_kernel(const uint4 *k, int64_t n, int lowbit, int nbits, unsigned long long *out)
{ unsigned long long bad = 0, s0 = 0, s1 = 0;
  const unsigned __int128 mask = (nbits >= 128 ? ~(unsigned __int128) 0 : ((((unsigned __int128) 1) << nbits) - 1));
  for (int64_t i = blockIdx.x*(int64_t) blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x*blockDim.x)
    { const uint4 c = k[i];
      const unsigned __int128 cv = ((unsigned __int128) c.w << 96) | ((unsigned __int128) c.z << 64) | ((unsigned __int128) c.y << 32) | c.x;
      s0 += mix((uint64_t) cv); s1 += mix((uint64_t) (cv >> 64) + 1);
      if (i > 0)
        { const uint4 p = k[i-1];
          const unsigned __int128 pv = ((unsigned __int128) p.w << 96) | ((unsigned __int128) p.z << 64) | ((unsigned __int128) p.y << 32) | p.x;
          if (((pv >> lowbit) & mask) > ((cv >> lowbit) & mask))
            bad += 1;
        }
    }
  atomicAdd(out,bad); atomicAdd(out+1,s0); atomicAdd(out+2,s1);
}

// the ceiling of a pass: the tile structure of the sort (a wavefront reads KPT rounds of 64 consecutive 16-byte records into
// registers, then writes them) with nothing in between
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int KPT, int NTM>
__global__ __launch_bounds__(256) void copy_kernel(const v4u *in, v4u *out, int64_t n)
{ const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = ((int64_t) blockIdx.x*4 + wave) * (64*KPT);
  v4u k[KPT];
  #pragma unroll
  for (int r = 0; r < KPT; r++)
    { const int64_t i = base + r*64 + lane;
      if (i < n) k[r] = NTM ? __builtin_nontemporal_load(in + i) : in[i];
    }
  #pragma unroll
  for (int r = 0; r < KPT; r++)
    { const int64_t i = base + r*64 + lane;
      if (i < n) { if (NTM) __builtin_nontemporal_store(k[r],out + i); else out[i] = k[r]; }
    }
}

template <int KPT, int NTM>
static void copy_bench(const uint4 *in, uint4 *out, int64_t n)
{ hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (int) ((n + 256*KPT - 1) / (256*KPT));
  float best = 1e30f;
  for (int r = 0; r < 4; r++)
    { CK(hipEventRecord(e0,0));
      hipLaunchKernelGGL((copy_kernel<KPT,NTM>),dim3(grid),dim3(256),0,0,(const v4u *) in,(v4u *) out,n);
      CK(hipEventRecord(e1,0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      if (ms < best) best = ms;
    }
  printf("copy: %d records per thread, %s: %.3f ms = %.2f TB/s (read + write)\n",KPT,NTM ? "nontemporal" : "plain",best,2.0*16.0*n/best*1e-9);
}

// the write pattern of a radix pass without the sort: tile t reads 4096 consecutive records and writes RUN-record runs, run f
// of tile t to front f at slot t (fronts advance tile by tile, neighbouring tiles share the cache lines at run boundaries when
// off != 0).  xcdlocal: tiles whose runs are neighbours run on the same XCD (block b -> tile (b%8)*(G/8) + b/8).
template <int RUN, int ST>
__global__ __launch_bounds__(256) void front_kernel(const v4u *in, v4u *out, int ntiles, int xcdlocal, int off, int transposed)
{ const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int t = xcdlocal ? (b & 7)*(ntiles >> 3) + (b >> 3) : b;
  const int64_t base = (int64_t) t*4096 + wave*1024;
  constexpr int NF = 4096/RUN;
  v4u k[16];
  #pragma unroll
  for (int r = 0; r < 16; r++)
    k[r] = in[base + r*64 + lane];
  #pragma unroll
  for (int r = 0; r < 16; r++)
    { const int j = transposed ? wave*1024 + lane*16 + r : wave*1024 + r*64 + lane, f = j / RUN;
      v4u *q = out + ((int64_t) f*ntiles*RUN + (int64_t) t*RUN + (j % RUN) + off);
      if (ST == 0) *q = k[r];
      else if (ST == 1) __builtin_nontemporal_store(k[r],q);
      else { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(k[r]) : "memory"); }
    }
  (void) NF;
}

template <int RUN, int ST>
static void front_bench(const uint4 *in, uint4 *out, int64_t n)
{ hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ntiles = (int) ((n / 4096 - 1) & ~7ll);
  for (int xl = 0; xl < 2; xl++)
  for (int tr = 0; tr < 2; tr++)
    for (int off = 0; off < 4; off += 3)
      { float best = 1e30f;
        for (int r = 0; r < 3; r++)
          { CK(hipEventRecord(e0,0));
            hipLaunchKernelGGL((front_kernel<RUN,ST>),dim3(ntiles),dim3(256),0,0,(const v4u *) in,(v4u *) out,ntiles,xl,off,tr);
            CK(hipEventRecord(e1,0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms,e0,e1));
            if (ms < best) best = ms;
          }
        printf("fronts%s: runs of %d records, store %s, %s, offset %d records: %.3f ms = %.2f TB/s (read + write)\n",
               xl ? " (neighbouring tiles on one XCD)" : "",RUN,
               ST == 0 ? "plain" : ST == 1 ? "nontemporal" : "sc0 sc1",
               tr ? "a block's 4 records from 4 instructions" : "a block from 4 lanes of one instruction",off,best,2.0*16.0*ntiles*4096/best*1e-9);
      }
}

typedef struct fga_dev fga_dev;

int main(int argc, char **argv)
{ if (argc < 6)
    { fprintf(stderr,"usage: sort_bench <n> <nbits> <lowbit> <uniform|seeds> <reps>\n"); return 2; }
  const int64_t n = atoll(argv[1]);
  const int nbits = atoi(argv[2]), lowbit = atoi(argv[3]);
  const int dist = strcmp(argv[4],"seeds") == 0;
  const int reps = atoi(argv[5]);
  const char *lib = getenv("FGA_LIBRARY");
  void *h = dlopen(lib != NULL ? lib : "fastga_amd/libfastga_amd.so",RTLD_NOW);
  if (h == NULL) { fprintf(stderr,"dlopen: %s\n",dlerror()); return 1; }
  int (*dev_open)(int, fga_dev **) = (int (*)(int, fga_dev **)) dlsym(h,"fga_dev_open");
  int (*sortf)(fga_dev *, void *, void *, int64_t, int, int, void **) =
      (int (*)(fga_dev *, void *, void *, int64_t, int, int, void **)) dlsym(h,"fga_dev_radix_sort_u128");
  float (*stage_ms)(const fga_dev *, int) = (float (*)(const fga_dev *, int)) dlsym(h,"fga_dev_stage_ms");
  const char *(*last_error)(void) = (const char *(*)(void)) dlsym(h,"fga_last_error");
  if (dev_open == NULL || sortf == NULL || stage_ms == NULL) { fprintf(stderr,"symbols missing\n"); return 1; }
  fga_dev *dev = NULL;
  if (dev_open(0,&dev)) { fprintf(stderr,"fga_dev_open: %s\n",last_error ? last_error() : "?"); return 1; }
  uint4 *orig, *b0, *b1;
  unsigned long long *chk, hc[6];
  CK(hipMalloc(&orig,sizeof(uint4)*(size_t) n)); CK(hipMalloc(&b0,sizeof(uint4)*(size_t) n)); CK(hipMalloc(&b1,sizeof(uint4)*(size_t) n));
  CK(hipMalloc(&chk,6*sizeof(unsigned long long)));
  CK(hipMemset(chk,0,6*sizeof(unsigned long long)));
  hipLaunchKernelGGL(fill_kernel,dim3(4096),dim3(256),0,0,orig,n,lowbit+nbits,dist,0x9e3779b97f4a7c15ull);
  hipLaunchKernelGGL(check_kernel,dim3(4096),dim3(256),0,0,orig,n,lowbit,nbits,chk);
  CK(hipDeviceSynchronize());
  if (getenv("SORT_BENCH_COPY") != NULL)
    { copy_bench<16,0>(orig,b0,n); copy_bench<8,0>(orig,b0,n); copy_bench<4,0>(orig,b0,n);
      copy_bench<16,1>(orig,b0,n); copy_bench<8,1>(orig,b0,n);
      front_bench<16,0>(orig,b0,n); front_bench<32,0>(orig,b0,n); front_bench<64,0>(orig,b0,n);
      return 0;
    }
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; r++)
    { CK(hipMemcpy(b0,orig,sizeof(uint4)*(size_t) n,hipMemcpyDeviceToDevice));
      CK(hipDeviceSynchronize());
      void *res = NULL;
      if (sortf(dev,b0,b1,n,lowbit,nbits,&res)) { fprintf(stderr,"sort failed: %s\n",last_error ? last_error() : "?"); return 1; }
      const float ms = stage_ms(dev,2);
      if (ms < best) best = ms;
      sum += ms;
      if (r == reps-1)
        { hipLaunchKernelGGL(check_kernel,dim3(4096),dim3(256),0,0,(const uint4 *) res,n,lowbit,nbits,chk+3);
          CK(hipDeviceSynchronize());
        }
    }
  CK(hipMemcpy(hc,chk,sizeof(hc),hipMemcpyDeviceToHost));
  const int ok = hc[3] == 0 && hc[1] == hc[4] && hc[2] == hc[5];
  const char *rbs = getenv("FGA_SORT_RB");
  const int rb = rbs != NULL ? atoi(rbs) : 8;
  const int npass = (nbits + rb - 1) / rb;
  printf("n=%lld nbits=%d lowbit=%d dist=%s  best %.3f ms  mean %.3f ms  = %.3f ms/pass (%d passes of %d bits)  %.2f TB/s per pass (2n16)  %s (unsorted pairs in input %llu, in output %llu)\n",
         (long long) n,nbits,lowbit,argv[4],best,sum/reps,best/npass,npass,rb,2.0*16.0*n*npass/best*1e-9,
         ok ? "SORTED+CHECKSUM OK" : "WRONG",hc[0],hc[3]);
  return ok ? 0 : 1;
}
