// tools/ubench/lone_wave.hip -- what a single resident wavefront pays per construct on gfx950 (the extension kernel's
// latency regime is one wavefront per SIMD running a long serial loop).  hipcc --offload-arch=gfx950 -O3; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define BARRIER(v) asm volatile("" : "+v"(v))

// scalar conditions in kernel arguments (SGPRs): s_cmp + s_cbranch around one VALU op
__global__ __launch_bounds__(64) void k_sbranch(int n, int *out, unsigned long long *cyc, int p0, int p1, int p2, int p3)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 10; j++)
        { if (p0 > i) { a += 3; BARRIER(a); }
          if (p1 > i) { a += 5; BARRIER(a); }
          if (p2 > i) { a += 7; BARRIER(a); }
          if (p3 > i) { a += 9; BARRIER(a); }
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

// the same work without branches: scalar select of the addend
__global__ __launch_bounds__(64) void k_sselect(int n, int *out, unsigned long long *cyc, int p0, int p1, int p2, int p3)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 10; j++)
        { a += (p0 > i) ? 3 : 0; BARRIER(a);
          a += (p1 > i) ? 5 : 0; BARRIER(a);
          a += (p2 > i) ? 7 : 0; BARRIER(a);
          a += (p3 > i) ? 9 : 0; BARRIER(a);
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

// condition from the lanes: v_cmp -> ballot -> scalar branch
__global__ __launch_bounds__(64) void k_ballot_branch(int n, int *out, unsigned long long *cyc, int thr)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 20; j++)
        { if (__builtin_amdgcn_ballot_w64(a > thr) != 0) { a += 3; BARRIER(a); }
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

// divergent condition: exec-mask region
__global__ __launch_bounds__(64) void k_exec_branch(int n, int *out, unsigned long long *cyc, int thr)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 20; j++)
        { if (a > thr) { a += 3; BARRIER(a); a ^= 1; BARRIER(a); a += 5; BARRIER(a); a ^= 2; BARRIER(a); a += 1; BARRIER(a); a ^= 3; BARRIER(a); }
          BARRIER(a);
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_valu_dep(int n, int *out, unsigned long long *cyc)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 50; j++)
        { a += 3; BARRIER(a); a ^= 5; BARRIER(a); }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_lds_chain(int n, int *out, unsigned long long *cyc)
{ __shared__ int tab[256];
  for (int i = threadIdx.x; i < 256; i += 64) tab[i] = (i*37 + 11) & 255;
  __syncthreads();
  int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 20; j++)
        a = tab[a & 255];
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_readlane_chain(int n, int *out, unsigned long long *cyc)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 20; j++)
        { const unsigned long long m = __builtin_amdgcn_ballot_w64(a > j);
          const int l = 63 - __builtin_clzll((long long) (m | 1));
          a += __builtin_amdgcn_readlane(a,l);
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_dpp_chain(int n, int *out, unsigned long long *cyc)
{ int a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 20; j++)
        { const int t = __builtin_amdgcn_update_dpp(0,a,0x138,0xf,0xf,false);
          a = a > t ? a+1 : t+2;
        }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

static unsigned long long *cyc; static int *out;
static double get(int n, int per)
{ unsigned long long h; hipDeviceSynchronize(); hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost); return (double) h/n/per; }

int main()
{ hipMalloc(&out,256*4); hipMalloc(&cyc,64);
  const int n = 20000, big = 1 << 30;
  hipLaunchKernelGGL(k_valu_dep,dim3(1),dim3(64),0,0,n,out,cyc);
  printf("dependent VALU op:                      %6.1f cycles\n",get(n,100));
  hipLaunchKernelGGL(k_sbranch,dim3(1),dim3(64),0,0,n,out,cyc,big,big,big,big);
  printf("s_cmp + s_cbranch (not taken) + 1 VALU: %6.1f cycles\n",get(n,40));
  hipLaunchKernelGGL(k_sbranch,dim3(1),dim3(64),0,0,n,out,cyc,0,0,0,0);
  printf("s_cmp + s_cbranch (taken), VALU skipped: %5.1f cycles\n",get(n,40));
  hipLaunchKernelGGL(k_sbranch,dim3(1),dim3(64),0,0,n,out,cyc,big,0,big,0);
  printf("the same, alternating:                  %6.1f cycles\n",get(n,40));
  hipLaunchKernelGGL(k_sselect,dim3(1),dim3(64),0,0,n,out,cyc,big,0,big,0);
  printf("s_cmp + s_cselect + 1 VALU (no branch): %6.1f cycles\n",get(n,40));
  hipLaunchKernelGGL(k_ballot_branch,dim3(1),dim3(64),0,0,n,out,cyc,-1);
  printf("v_cmp -> ballot -> branch (not taken) + VALU: %4.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_ballot_branch,dim3(1),dim3(64),0,0,n,out,cyc,big);
  printf("v_cmp -> ballot -> branch (taken):      %6.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_exec_branch,dim3(1),dim3(64),0,0,n,out,cyc,-1);
  printf("divergent if, 6 VALU inside, all lanes in: %5.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_exec_branch,dim3(1),dim3(64),0,0,n,out,cyc,big);
  printf("divergent if, no lane in (execz skip):  %6.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_lds_chain,dim3(1),dim3(64),0,0,n,out,cyc);
  printf("dependent LDS read:                     %6.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_readlane_chain,dim3(1),dim3(64),0,0,n,out,cyc);
  printf("ballot -> flbit -> readlane -> add:     %6.1f cycles\n",get(n,20));
  hipLaunchKernelGGL(k_dpp_chain,dim3(1),dim3(64),0,0,n,out,cyc);
  printf("dpp mov + cmp + select:                 %6.1f cycles\n",get(n,20));
  return 0;
}
