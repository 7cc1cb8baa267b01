// round 6 probe 2: what a hipMalloc costs as a function of what the process holds already, after frees, after a pause, and
// through the virtual-memory calls.  hipcc --offload-arch=gfx950 -O2 -o tools/probes/_probe_alloc2 tools/probes/r6_probe_alloc2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <chrono>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4 *p, size_t n)
{ size_t i = blockIdx.x*(size_t) blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t) gridDim.x*blockDim.x) { uint4 v = p[i]; v.x += 1; p[i] = v; }
}
int main(int argc, char **argv)
{ const size_t GB = (size_t) 1 << 30;
  const size_t chunk = (argc > 1 ? atol(argv[1]) : 8) * GB;
  const int n = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<void *> p((size_t) n,NULL);
  size_t fr, tot; hipMemGetInfo(&fr,&tot); printf("free %.1f GB of %.1f\n",fr/1e9,tot/1e9);
  printf("A: %d x hipMalloc %zu GB one after the other:",n,chunk/GB);
  for (int k = 0; k < n; k++) { double t = now(); hipError_t e = hipMalloc(&p[k],chunk); printf(" %.0f%s",1e3*(now()-t),e == hipSuccess ? "" : "!"); }
  printf(" ms\n");
  { double t = now(); for (int k = 0; k < n; k++) hipFree(p[k]); printf("B: all freed in %.0f ms\n",1e3*(now()-t)); }
  printf("C: again at once:");
  for (int k = 0; k < n; k++) { double t = now(); hipMalloc(&p[k],chunk); printf(" %.0f",1e3*(now()-t)); }
  printf(" ms\n");
  for (int k = 0; k < n; k++) hipFree(p[k]);
  sleep(5);
  printf("D: again after 5 s:");
  for (int k = 0; k < n; k++) { double t = now(); hipMalloc(&p[k],chunk); printf(" %.0f",1e3*(now()-t)); }
  printf(" ms\n");
  { hipStream_t s; hipStreamCreate(&s); double t = now();
    hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,(uint4 *) p[n-1],chunk/16); hipStreamSynchronize(s);
    printf("   first kernel over the last chunk %.1f ms,",1e3*(now()-t)); t = now();
    hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,(uint4 *) p[n-1],chunk/16); hipStreamSynchronize(s);
    printf(" second %.1f ms\n",1e3*(now()-t));
  }
  for (int k = 0; k < n; k++) hipFree(p[k]);
  // one big region
  { void *q; double t = now(); hipError_t e = hipMalloc(&q,chunk*n); printf("E: one hipMalloc of %zu GB: %.0f ms (%s)\n",chunk*n/GB,1e3*(now()-t),hipGetErrorString(e)); hipFree(q); }
  // vmm
  { hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    void *va = NULL; hipError_t e = hipMemAddressReserve(&va,chunk*n,0,NULL,0);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    printf("F: vmm create+map+access per %zu GB chunk:",chunk/GB);
    std::vector<hipMemGenericAllocationHandle_t> h((size_t) n);
    for (int k = 0; k < n && e == hipSuccess; k++)
      { double t = now();
        e = hipMemCreate(&h[k],chunk,&prop,0);
        if (e == hipSuccess) e = hipMemMap((char *) va + chunk*k,chunk,0,h[k],0);
        if (e == hipSuccess) e = hipMemSetAccess((char *) va + chunk*k,chunk,&acc,1);
        printf(" %.0f",1e3*(now()-t));
      }
    printf(" ms (%s)\n",hipGetErrorString(e));
    hipStream_t s; hipStreamCreate(&s); double t = now();
    hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,(uint4 *) va,chunk*n/16); e = hipStreamSynchronize(s);
    printf("   kernel over all of it %.1f ms (%s)\n",1e3*(now()-t),hipGetErrorString(e));
  }
  return 0;
}
