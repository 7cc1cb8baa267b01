// tools/ubench/alloc_probe.hip -- what device memory costs to obtain on this box: hipMalloc / first touch / hipFree of large
// pieces, a second round in the same process, and the virtual-memory API (hipMemCreate + hipMemMap) for comparison.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/alloc_probe.hip -o fastga_amd/bin/alloc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/time.h>
#include <thread>
#include <vector>
static double now() { struct timeval t; gettimeofday(&t,NULL); return t.tv_sec + 1e-6*t.tv_usec; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n",#x,hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv)
{ const size_t GB = (size_t) 1 << 30;
  const size_t gb = argc > 1 ? (size_t) atoll(argv[1]) : 48;
  double t = now();
  CK(hipSetDevice(0)); CK(hipFree(0));
  printf("context: %.3f s\n",now()-t);
  const int mode = argc > 2 ? atoi(argv[2]) : 0;      // 1: virtual-memory API first, 2: hipMallocAsync first, 3: hipHostMalloc-free path
  if (mode == 1)
    { hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      for (int i = 0; i < 3; i++)
        { const size_t sz = gb*GB;
          void *va; hipMemGenericAllocationHandle_t h;
          t = now(); CK(hipMemAddressReserve(&va,sz,0,0,0)); CK(hipMemCreate(&h,sz,&prop,0)); CK(hipMemMap(va,sz,0,h,0)); CK(hipMemSetAccess(va,sz,&acc,1));
          printf("FIRST: VMM create+map %zu GB: %.3f s\n",gb,now()-t);
          t = now(); CK(hipMemset(va,1,sz)); CK(hipDeviceSynchronize()); printf("FIRST: memset: %.3f s\n",now()-t);
          t = now(); CK(hipMemset(va,1,sz)); CK(hipDeviceSynchronize()); printf("FIRST: memset again: %.3f s\n",now()-t);
        }
    }
  if (mode == 3)                                       // concurrent hipMalloc from several host threads
    { const int nth = argc > 3 ? atoi(argv[3]) : 4;
      std::vector<std::thread> th;
      std::vector<void *> ptr((size_t) nth,(void *) NULL);
      t = now();
      for (int i = 0; i < nth; i++)
        th.emplace_back([&,i]() { hipSetDevice(0); double t0 = now(); hipError_t e = hipMalloc(&ptr[(size_t) i],gb*GB);
                                  printf("  thread %d: hipMalloc %zu GB %s in %.3f s\n",i,gb,hipGetErrorString(e),now()-t0); });
      for (auto &x : th) x.join();
      printf("FIRST: %d threads x %zu GB concurrently: %.3f s\n",nth,gb,now()-t);
      t = now(); CK(hipMemset(ptr[0],1,gb*GB)); CK(hipDeviceSynchronize()); printf("FIRST: memset: %.3f s\n",now()-t);
      return 0;
    }
  if (mode == 4)                                       // one contiguous range made of chunks created concurrently (virtual-memory API)
    { const int nth = argc > 3 ? atoi(argv[3]) : 16;
      const size_t chunk = gb*GB;
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      for (int wave = 0; wave < 2; wave++)
        { void *va = NULL;
          t = now();
          CK(hipMemAddressReserve(&va,chunk*nth,0,0,0));
          std::vector<std::thread> th;
          for (int i = 0; i < nth; i++)
            th.emplace_back([&,i]() { hipSetDevice(0); double t0 = now(); hipMemGenericAllocationHandle_t h;
                                      hipError_t e = hipMemCreate(&h,chunk,&prop,0);
                                      if (e == hipSuccess) e = hipMemMap((char *) va + (size_t) i*chunk,chunk,0,h,0);
                                      if (now()-t0 > 0.05 || e != hipSuccess) printf("  chunk %d: %s in %.3f s\n",i,hipGetErrorString(e),now()-t0); });
          for (auto &x : th) x.join();
          CK(hipMemSetAccess(va,chunk*nth,&acc,1));
          printf("FIRST: wave %d: %d chunks x %zu GB created concurrently, mapped as one range: %.3f s\n",wave,nth,gb,now()-t);
          t = now(); CK(hipMemset(va,1,chunk*nth)); CK(hipDeviceSynchronize()); printf("FIRST: memset of the range: %.3f s\n",now()-t);
        }
      return 0;
    }
  if (mode == 2)
    for (int i = 0; i < 3; i++)
      { void *q; t = now(); CK(hipMallocAsync(&q,gb*GB,0)); CK(hipStreamSynchronize(0)); printf("FIRST: hipMallocAsync %zu GB: %.3f s\n",gb,now()-t);
        t = now(); CK(hipMemset(q,1,gb*GB)); CK(hipDeviceSynchronize()); printf("FIRST: memset: %.3f s\n",now()-t);
      }
  for (int round = 0; round < 2; round++)
    { void *p[3];
      for (int i = 0; i < 3; i++)
        { t = now(); CK(hipMalloc(&p[i],gb*GB)); printf("round %d hipMalloc %zu GB: %.3f s\n",round,gb,now()-t); }
      t = now(); CK(hipMemset(p[0],1,gb*GB)); CK(hipDeviceSynchronize()); printf("round %d first memset of piece 0: %.3f s\n",round,now()-t);
      t = now(); CK(hipMemset(p[0],2,gb*GB)); CK(hipDeviceSynchronize()); printf("round %d second memset: %.3f s\n",round,now()-t);
      for (int i = 0; i < 3; i++)
        { t = now(); CK(hipFree(p[i])); printf("round %d hipFree: %.3f s\n",round,now()-t); }
    }
  { // one big piece
    void *q; t = now(); CK(hipMalloc(&q,3*gb*GB)); printf("hipMalloc %zu GB in one piece: %.3f s\n",3*gb,now()-t);
    t = now(); CK(hipFree(q)); printf("hipFree: %.3f s\n",now()-t);
  }
  { // virtual memory API
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran,&prop,hipMemAllocationGranularityRecommended));
    printf("VMM granularity %zu\n",gran);
    const size_t sz = gb*GB;
    void *va; t = now(); CK(hipMemAddressReserve(&va,sz,0,0,0)); printf("reserve: %.3f s\n",now()-t);
    hipMemGenericAllocationHandle_t h; t = now(); CK(hipMemCreate(&h,sz,&prop,0)); printf("hipMemCreate %zu GB: %.3f s\n",gb,now()-t);
    t = now(); CK(hipMemMap(va,sz,0,h,0)); printf("hipMemMap: %.3f s\n",now()-t);
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    t = now(); CK(hipMemSetAccess(va,sz,&acc,1)); printf("hipMemSetAccess: %.3f s\n",now()-t);
    t = now(); CK(hipMemset(va,1,sz)); CK(hipDeviceSynchronize()); printf("first memset: %.3f s\n",now()-t);
    t = now(); CK(hipMemUnmap(va,sz)); CK(hipMemRelease(h)); CK(hipMemAddressFree(va,sz)); printf("unmap+release: %.3f s\n",now()-t);
  }
  { void *q; t = now(); CK(hipMallocAsync(&q,gb*GB,0)); CK(hipStreamSynchronize(0)); printf("hipMallocAsync %zu GB: %.3f s\n",gb,now()-t);
    t = now(); CK(hipFreeAsync(q,0)); CK(hipStreamSynchronize(0)); printf("hipFreeAsync: %.3f s\n",now()-t);
    t = now(); CK(hipMallocAsync(&q,gb*GB,0)); CK(hipStreamSynchronize(0)); printf("hipMallocAsync again: %.3f s\n",now()-t);
  }
  return 0;
}
