// round 6 probe: does a hipMalloc of tens of GB on one thread (the driver clears the pages: ~25 ms per GB) run beside kernels
// and copies launched from another thread?  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_alloc tools/probes/r6_probe_alloc.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <atomic>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4 *p, size_t n)
{ size_t i = blockIdx.x*(size_t) blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t) gridDim.x*blockDim.x) { uint4 v = p[i]; v.x += 1; p[i] = v; }
}
int main(int argc, char **argv)
{ const size_t GB = (size_t) 1 << 30;
  size_t big = (argc > 1 ? atol(argv[1]) : 32) * GB;
  uint4 *w; hipMalloc(&w,4*GB); hipMemset(w,0,4*GB); hipDeviceSynchronize();
  hipStream_t s; hipStreamCreate(&s);
  // baseline: kernel time alone
  double t0 = now();
  for (int k = 0; k < 50; k++) hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,w,4*GB/16);
  hipStreamSynchronize(s);
  double alone = now() - t0;
  // baseline: malloc alone
  void *a = NULL; t0 = now(); hipMalloc(&a,big); double malone = now() - t0;
  printf("50 kernels alone %.3f s   hipMalloc %zu GB alone %.3f s\n",alone,big/GB,malone);
  // together
  std::atomic<int> done(0); double mtime = 0; void *b = NULL;
  std::thread th([&]{ hipSetDevice(0); double t = now(); hipMalloc(&b,big); mtime = now() - t; done = 1; });
  t0 = now(); int launched = 0; double worst = 0;
  while (!done || launched < 50)
    { double t1 = now();
      hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,w,4*GB/16);
      hipStreamSynchronize(s);
      double d = now() - t1; if (d > worst) worst = d;
      launched += 1;
    }
  double both = now() - t0;
  th.join();
  printf("together: %d kernels in %.3f s (%.4f s each, worst %.4f; alone %.4f each), hipMalloc beside them %.3f s\n",
         launched,both,both/launched,worst,alone/50,mtime);
  // freed and taken again: does the driver clear again?
  hipFree(a); t0 = now(); hipMalloc(&a,big); printf("free + hipMalloc again %.3f s\n",now()-t0);
  // virtual memory management: reserve, create + map in 2 GB granules
  { hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; hipError_t e = hipMemGetAllocationGranularity(&gran,&prop,hipMemAllocationGranularityRecommended);
    printf("vmm granularity %zu (%s)\n",gran,hipGetErrorString(e));
    hipFree(a); hipFree(b);
    void *va = NULL; e = hipMemAddressReserve(&va,big,0,NULL,0); printf("reserve: %s\n",hipGetErrorString(e));
    const size_t chunk = 2*GB; double tc = 0, tm = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t off = 0; off < big && e == hipSuccess; off += chunk)
      { hipMemGenericAllocationHandle_t h; double t = now();
        e = hipMemCreate(&h,chunk,&prop,0); tc += now()-t; t = now();
        if (e == hipSuccess) e = hipMemMap((char *) va + off,chunk,0,h,0);
        if (e == hipSuccess) e = hipMemSetAccess((char *) va + off,chunk,&acc,1);
        tm += now()-t;
      }
    printf("vmm: create %.3f s, map+access %.3f s for %zu GB (%s)\n",tc,tm,big/GB,hipGetErrorString(e));
    if (e == hipSuccess)
      { t0 = now(); hipLaunchKernelGGL(touch,dim3(4096),dim3(256),0,s,(uint4 *) va,big/16); e = hipStreamSynchronize(s);
        printf("kernel over the mapped range: %.3f s (%s)\n",now()-t0,hipGetErrorString(e)); }
  }
  return 0;
}
