// tools/ubench/ilp_probe.hip -- does the compiler interleave two independent dependency chains of one basic block for a
// lone wavefront (in-order issue, 8 cycles per dependent VALU op)?  Chain A then chain B in source order.
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEP(v) { v ^= v >> 3; v += 0x9e3779b9u; v ^= v << 5; v += 0x7f4a7c15u; }

__global__ __launch_bounds__(64) void k_one(int n, unsigned *out, unsigned long long *cyc)
{ unsigned a = threadIdx.x;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 25; j++) STEP(a)
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_two(int n, unsigned *out, unsigned long long *cyc)
{ unsigned a = threadIdx.x, b = threadIdx.x*7+1;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 25; j++) STEP(a)
#pragma unroll
      for (int j = 0; j < 25; j++) STEP(b)
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a+b;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

__global__ __launch_bounds__(64) void k_two_hand(int n, unsigned *out, unsigned long long *cyc)
{ unsigned a = threadIdx.x, b = threadIdx.x*7+1;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; i++)
    {
#pragma unroll
      for (int j = 0; j < 25; j++) { STEP(a) STEP(b) }
    }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = a+b;
  if (threadIdx.x == 0) cyc[0] = t1-t0;
}

int main()
{ unsigned *out; unsigned long long *cyc, h;
  hipMalloc(&out,1024); hipMalloc(&cyc,64);
  const int n = 20000;
  hipLaunchKernelGGL(k_one,dim3(1),dim3(64),0,0,n,out,cyc); hipDeviceSynchronize(); hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost);
  printf("one chain of 100 dependent op groups:           %.1f cycles per iteration\n",(double) h/n);
  hipLaunchKernelGGL(k_two,dim3(1),dim3(64),0,0,n,out,cyc); hipDeviceSynchronize(); hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost);
  printf("two chains, one after the other in the source:  %.1f cycles per iteration\n",(double) h/n);
  hipLaunchKernelGGL(k_two_hand,dim3(1),dim3(64),0,0,n,out,cyc); hipDeviceSynchronize(); hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost);
  printf("two chains, interleaved by hand in the source:  %.1f cycles per iteration\n",(double) h/n);
  return 0;
}
