// tools/ubench/exec_probe.hip -- does a wave64 VALU instruction cost less when only the low 32 (16) lanes are active?
// 4 / 8 wavefronts per SIMD, each a long chain of independent v_fma / v_add_u32 / v_cndmask under an EXEC mask set once.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/exec_probe.hip -o fastga_amd/bin/exec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1; } } while (0)

template <int ACTIVE>
__global__ __launch_bounds__(256) void probe(float *out, int iters)
{ const int lane = threadIdx.x & 63;
  float a0 = lane, a1 = lane+1, a2 = lane+2, a3 = lane+3, a4 = lane+4, a5 = lane+5, a6 = lane+6, a7 = lane+7;
  unsigned u0 = lane, u1 = lane*3;
  if (lane < ACTIVE)                          // one exec region around the whole loop
    { for (int i = 0; i < iters; i++)
        { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                       "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                       "v_add_u32 %8, %8, %9\n v_xor_b32 %9, %9, %8\n v_add_u32 %8, %8, %9\n v_xor_b32 %9, %9, %8\n"
                       "v_add_u32 %8, %8, %9\n v_xor_b32 %9, %9, %8\n v_add_u32 %8, %8, %9\n v_xor_b32 %9, %9, %8\n"
                       : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7),"+v"(u0),"+v"(u1));
        }
    }
  out[blockIdx.x*256 + threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7 + (float) (u0 ^ u1);
}

template <int ACTIVE>
static int run(float *d, int wgs, int iters, const char *what)
{ hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 3; r++)
    { CK(hipEventRecord(e0,0));
      hipLaunchKernelGGL(probe<ACTIVE>,dim3(wgs),dim3(256),0,0,d,iters);
      CK(hipEventRecord(e1,0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      if (ms < best) best = ms;
    }
  const double insts = (double) wgs*4*iters*16;
  printf("%-28s %d workgroups: %.3f ms, %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n",what,wgs,best,
         best*1e-3*2.4e9 / (insts / 1024.0));
  return 0;
}

int main()
{ float *d; CK(hipMalloc(&d,sizeof(float)*256*8192));
  for (int wpc = 4; wpc <= 8; wpc += 4)          // workgroups per CU -> 4 / 8 waves per SIMD
    { const int wgs = 256*wpc, it = 20000;
      printf("-- %d wavefronts per SIMD\n",wpc);
      run<64>(d,wgs,it,"all 64 lanes"); run<32>(d,wgs,it,"lanes 0..31"); run<16>(d,wgs,it,"lanes 0..15"); run<48>(d,wgs,it,"lanes 0..47");
    }
  return 0;
}
