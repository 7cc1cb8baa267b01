#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *shl, int *shr)
{ int v = threadIdx.x;
  shl[threadIdx.x] = __builtin_amdgcn_update_dpp(-7,v,0x130,0xf,0xf,false);
  shr[threadIdx.x] = __builtin_amdgcn_update_dpp(-7,v,0x138,0xf,0xf,false);
}
int main()
{ int *a, *b, ha[64], hb[64];
  hipMalloc(&a,256); hipMalloc(&b,256);
  hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,a,b);
  hipMemcpy(ha,a,256,hipMemcpyDeviceToHost); hipMemcpy(hb,b,256,hipMemcpyDeviceToHost);
  printf("wave_shl:1 :"); for (int i = 0; i < 64; i++) printf(" %d",ha[i]); printf("\n");
  printf("wave_shr:1 :"); for (int i = 0; i < 64; i++) printf(" %d",hb[i]); printf("\n");
  return 0;
}
