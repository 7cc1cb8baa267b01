// tools/ubench/alloc_sizes.hip -- hipMalloc / hipFree time by allocation size in one process (sizes in GB on the command
// line, in the order given): is the cost per GB, or a step at some size?
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/alloc_sizes.hip -o fastga_amd/bin/alloc_sizes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
static double now() { struct timeval t; gettimeofday(&t,NULL); return t.tv_sec + 1e-6*t.tv_usec; }
int main(int argc, char **argv)
{ hipSetDevice(0); hipFree(0);
  const int keep = getenv("ALLOC_KEEP") != NULL;         // keep every piece (total must fit the device)
  for (int i = 1; i < argc; i++)
    { const double gb = atof(argv[i]);
      void *p = NULL;
      double t = now();
      hipError_t e = hipMalloc(&p,(size_t) (gb*1073741824.0));
      const double ta = now() - t;
      t = now();
      if (e == hipSuccess) { hipMemsetAsync(p,1,1 << 20,0); hipDeviceSynchronize(); }
      const double tt = now() - t;
      t = now();
      if (e == hipSuccess && !keep) hipFree(p);
      printf("%6.1f GB: hipMalloc %8.1f ms (%s), first touch of 1 MB %6.1f ms, hipFree %8.1f ms\n",gb,1e3*ta,hipGetErrorString(e),1e3*tt,
             keep ? 0. : 1e3*(now()-t));
      fflush(stdout);
    }
  return 0;
}
