// fga_device.hip -- device context, index upload, seed buffers (C-ABI of include/fastga_amd.h).
#include "fga_device.hpp"
#include <stdio.h>

extern "C" int fga_dev_open(int device, fga_dev **out)
{ *out = NULL;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    { fga_set_error("no HIP device available (%s): libfastga_amd has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
      return 1;
    }
  if (device < 0 || device >= n)
    { fga_set_error("device %d out of range (have %d)",device,n);
      return 1;
    }
  FGA_HIP(hipSetDevice(device));
  fga_dev *d = (fga_dev *) calloc(1,sizeof(fga_dev));
  if (d == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  d->device = device;
  hipDeviceProp_t prop;
  FGA_HIP(hipGetDeviceProperties(&prop,device));
  d->ncu = prop.multiProcessorCount;
  FGA_HIP(hipStreamCreateWithFlags(&d->stream,hipStreamNonBlocking));
  FGA_HIP(hipEventCreate(&d->ev0));
  FGA_HIP(hipEventCreate(&d->ev1));
  *out = d;
  return 0;
}

// idle workspace slots go back to the device (the slots are grow-only while in use; after an index build, or between the
// passes of a multi-pass run, tens of GB sit in slots nobody will ask for at that size again)
extern "C" void fga_dev_trim(fga_dev *dev)
{ for (int q = 0; q < SLOT_COUNT; q++)
    if (dev->slot_ptr[q] != NULL && !dev->slot_busy[q])
      { hipFree(dev->slot_ptr[q]);
        dev->slot_ptr[q] = NULL; dev->slot_bytes[q] = 0;
      }
}

// device memory an allocation could get right now: free memory + what the idle slots hold
extern "C" size_t fga_dev_available(fga_dev *dev)
{ size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr,&tot) != hipSuccess)
    return 0;
  for (int q = 0; q < SLOT_COUNT; q++)
    if (dev->slot_ptr[q] != NULL && !dev->slot_busy[q])
      fr += dev->slot_bytes[q];
  return fr;
}

// FGA_TIMING=1: wall-clock notes on stderr (allocations of a GB and more, the steps of an index build): where a
// human-scale run spends its host time
extern "C" void fga_note(const char *what, double since)
{ static int on = -1;
  if (on < 0) on = getenv("FGA_TIMING") != NULL && atoi(getenv("FGA_TIMING")) != 0;
  if (on)
    fprintf(stderr,"[fga timing] %-40s %9.1f ms\n",what,1e3*(fga_wall() - since));
}

static void *alloc_or_trim(fga_dev *dev, size_t bytes)
{ void *p = NULL;
  const double t0 = fga_wall();
  if (hipMalloc(&p,bytes) == hipSuccess)
    { if (bytes >= ((size_t) 1 << 30))
        { char what[64];
          snprintf(what,sizeof(what),"hipMalloc %.1f GB",bytes*1e-9);
          fga_note(what,t0);
        }
      return p;
    }
  (void) hipGetLastError();
  fga_dev_trim(dev);                        // give the idle slots back and try once more
  if (hipMalloc(&p,bytes) == hipSuccess)
    return p;
  (void) hipGetLastError();
  return NULL;
}

void *fga_dev_acquire(fga_dev *dev, int slot, size_t bytes)
{ if (bytes == 0) bytes = 16;
  if (slot < 0 || slot >= SLOT_COUNT || dev->slot_busy[slot])
    return alloc_or_trim(dev,bytes);        // slot taken (or none asked): a private allocation
  if (dev->slot_ptr[slot] == NULL || dev->slot_bytes[slot] < bytes)
    { // another idle slot may hold a buffer that is large enough (the index builder's key buffers, the undivided seed
      // buffer of a multi-pass run): take it over instead of freeing / allocating tens of GB -- a hipMalloc that follows
      // the hipFree of such a buffer has been measured at 1-3 s
      int best = -1;
      for (int q = 0; q < SLOT_COUNT; q++)
        if (q != slot && !dev->slot_busy[q] && dev->slot_ptr[q] != NULL && dev->slot_bytes[q] >= bytes &&
            (best < 0 || dev->slot_bytes[q] < dev->slot_bytes[best]))
          best = q;
      size_t fr = 0, tot = 0;
      if (best >= 0 && bytes >= ((size_t) 64 << 20) && dev->slot_bytes[best] > 2*bytes + ((size_t) 1 << 30) &&
          hipMemGetInfo(&fr,&tot) == hipSuccess && fr < ((size_t) 32 << 30))
        { // far larger than what is asked for (a part's buffers after the undivided ones) and the device is nearly full:
          // give it back and allocate what is needed.  With room to spare the large buffer is taken over all the same: a
          // fresh allocation of tens of GB after such a free has been measured at 3 s (the extension's 64 GB trace-point
          // pool at 10 % divergence)
          const double t0 = fga_wall();
          hipFree(dev->slot_ptr[best]);
          fga_note("hipFree of an oversize idle slot",t0);
          dev->slot_ptr[best] = NULL; dev->slot_bytes[best] = 0;
          best = -1;
        }
      if (best >= 0 && bytes >= ((size_t) 64 << 20))
        { void *tp = dev->slot_ptr[slot]; size_t tb = dev->slot_bytes[slot];
          dev->slot_ptr[slot] = dev->slot_ptr[best]; dev->slot_bytes[slot] = dev->slot_bytes[best];
          dev->slot_ptr[best] = tp; dev->slot_bytes[best] = tb;
        }
      else
        { if (dev->slot_ptr[slot] != NULL)
            hipFree(dev->slot_ptr[slot]);
          dev->slot_ptr[slot] = NULL; dev->slot_bytes[slot] = 0;
          size_t slack = bytes/8;               // room to grow without a new allocation, bounded: a 50 GB buffer does not get 6 GB of it
          if (slack > ((size_t) 256 << 20)) slack = (size_t) 256 << 20;
          size_t want = bytes + slack;
          dev->slot_ptr[slot] = alloc_or_trim(dev,want);
          if (dev->slot_ptr[slot] == NULL)
            return NULL;
          dev->slot_bytes[slot] = want;
        }
    }
  dev->slot_busy[slot] = 1;
  return dev->slot_ptr[slot];
}

void fga_dev_release(fga_dev *dev, int slot, void *ptr)
{ if (ptr == NULL) return;
  if (slot >= 0 && slot < SLOT_COUNT && dev->slot_ptr[slot] == ptr)
    dev->slot_busy[slot] = 0;
  else
    hipFree(ptr);
}

void *fga_dev_pinned(fga_dev *dev, size_t bytes)
{ if (dev->pinned == NULL || dev->pinned_bytes < bytes)
    { if (dev->pinned != NULL) hipHostFree(dev->pinned);
      dev->pinned = NULL;
      if (hipHostMalloc(&dev->pinned,bytes + bytes/8,hipHostMallocDefault) != hipSuccess)
        { dev->pinned = NULL; dev->pinned_bytes = 0;
          return NULL;
        }
      dev->pinned_bytes = bytes + bytes/8;
    }
  return dev->pinned;
}

extern "C" void *fga_dev_stage_acquire(fga_dev *dev, size_t bytes)
{ if (hipSetDevice(dev->device) != hipSuccess) return NULL;
  void *p = fga_dev_acquire(dev,SLOT_STAGE,bytes);
  if (p == NULL)
    fga_set_error("device allocation of %zu bytes (seed staging) failed",bytes);
  return p;
}

extern "C" void fga_dev_stage_release(fga_dev *dev, void *ptr)
{ fga_dev_release(dev,SLOT_STAGE,ptr); }

extern "C" int fga_dev_malloc(fga_dev *dev, size_t bytes, void **out)
{ *out = NULL;
  FGA_HIP(hipSetDevice(dev->device));
  *out = alloc_or_trim(dev,bytes > 0 ? bytes : 16);        // idle workspace slots are given back before giving up
  if (*out == NULL)
    { fga_set_error("device allocation of %zu bytes failed: out of memory",bytes);
      return 1;
    }
  return 0;
}

extern "C" void fga_dev_free(fga_dev *dev, void *ptr)
{ if (ptr == NULL) return;
  hipSetDevice(dev->device);
  hipFree(ptr);
}

// free device memory right now, remembered as a low-water mark: with the grow-only workspace slots, the difference to
// the device's total is the peak footprint of the process on this GPU (fga_run_stats.hbm_peak_bytes)
void fga_dev_note_memory(fga_dev *dev)
{ size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr,&tot) == hipSuccess && (dev->hbm_low_water == 0 || fr < dev->hbm_low_water))
    dev->hbm_low_water = fr;
}

extern "C" void fga_dev_set_host_threads(fga_dev *dev, int nthreads)
{ dev->host_threads = nthreads > 0 ? nthreads : 1; }

extern "C" int64_t fga_dev_peak_bytes(fga_dev *dev)
{ size_t fr = 0, tot = 0;
  if (hipSetDevice(dev->device) != hipSuccess || hipMemGetInfo(&fr,&tot) != hipSuccess)
    return -1;
  fga_dev_note_memory(dev);
  return (int64_t) (tot - dev->hbm_low_water);
}

extern "C" int fga_dev_download(fga_dev *dev, void *host_dst, const void *device_src, size_t bytes)
{ FGA_HIP(hipSetDevice(dev->device));
  if (bytes > 0)
    FGA_HIP(hipMemcpy(host_dst,device_src,bytes,hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int fga_dev_upload(fga_dev *dev, void *device_dst, const void *host_src, size_t bytes)
{ FGA_HIP(hipSetDevice(dev->device));
  if (bytes > 0)
    FGA_HIP(hipMemcpy(device_dst,host_src,bytes,hipMemcpyHostToDevice));
  return 0;
}

extern "C" void fga_dev_close(fga_dev *d)
{ if (d == NULL) return;
  hipSetDevice(d->device);
  hipStreamSynchronize(d->stream);
  for (int q = 0; q < SLOT_COUNT; q++)
    if (d->slot_ptr[q] != NULL) hipFree(d->slot_ptr[q]);
  if (d->pinned != NULL) hipHostFree(d->pinned);
  hipEventDestroy(d->ev0);
  hipEventDestroy(d->ev1);
  hipStreamDestroy(d->stream);
  free(d);
}

extern "C" int fga_dev_sync(fga_dev *d)
{ FGA_HIP(hipSetDevice(d->device));
  FGA_HIP(hipStreamSynchronize(d->stream));
  return 0;
}

extern "C" float fga_dev_stage_ms(const fga_dev *d, int stage)
{ if (stage < 0 || stage >= FGA_NSTAGES) return -1.f;
  return d->last_ms[stage];
}

// the entries of the 12-mer prefixes [pbeg,pend) of a host-resident table (the whole table: 0, 2^24) on the device
static int dgix_upload_impl(fga_dev *dev, const fga_gix *X, int64_t pbeg, int64_t pend, fga_dgix **out)
{ *out = NULL;
  FGA_HIP(hipSetDevice(dev->device));
  if (X->table == NULL || X->index == NULL)
    { fga_set_error("fga_dgix_upload: the index holds no host copy of its table");
      return 1;
    }
  fga_dgix *D = (fga_dgix *) calloc(1,sizeof(fga_dgix));
  if (D == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  const bool whole = pbeg <= 0 && pend >= FGA_NPREFIX;
  const int64_t lo = (whole || pbeg <= 0) ? 0 : X->index[pbeg-1], hi = whole ? X->nents : X->index[pend-1];
  int64_t *sub = NULL;
  const int64_t *hidx = X->index;
  D->dev = dev;
  D->nents = hi - lo; D->ebytes = X->ebytes; D->postbytes = X->postbytes; D->contbytes = X->contbytes;
  D->nctg = X->nctg;
  if (!whole)                                       // the slice's own cumulative counts
    { sub = (int64_t *) malloc(sizeof(int64_t)*FGA_NPREFIX);
      if (sub == NULL)
        { fga_set_error("out of memory"); free(D);
          return 1;
        }
      for (int64_t p = 0; p < FGA_NPREFIX; p++)
        { const int64_t v = X->index[p];
          sub[p] = (v < lo ? lo : (v > hi ? hi : v)) - lo;
        }
      hidx = sub;
    }
  size_t tbytes = (size_t) D->nents * X->ebytes;
  hipError_t e;
  if ((e = hipMalloc(&D->table,tbytes + 64)) != hipSuccess ||
      (e = hipMalloc(&D->index,sizeof(int64_t)*FGA_NPREFIX)) != hipSuccess)
    { fga_set_error("fga_dgix_upload: device allocation of %zu bytes failed: %s",tbytes,hipGetErrorString(e));
      hipFree(D->table); hipFree(D->index); free(D); free(sub);
      return 1;
    }
  if ((e = hipMemcpy(D->table,X->table + (size_t) lo*X->ebytes,tbytes,hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemset(D->table + tbytes,0,64)) != hipSuccess ||
      (e = hipMemcpy(D->index,hidx,sizeof(int64_t)*FGA_NPREFIX,hipMemcpyHostToDevice)) != hipSuccess)
    { fga_set_error("fga_dgix_upload: copy failed: %s",hipGetErrorString(e));
      hipFree(D->table); hipFree(D->index); free(D); free(sub);
      return 1;
    }
  free(sub);
  D->legacy_cutoff = X->legacy ? X->freq : 0;
  if (fga_dgix_make_view(dev,D,0))
    { hipFree(D->table); hipFree(D->index); free(D);
      return 1;
    }
  *out = D;
  return 0;
}

extern "C" int fga_dgix_upload(fga_dev *dev, const fga_gix *X, fga_dgix **out)
{ return dgix_upload_impl(dev,X,0,FGA_NPREFIX,out); }

// one rank's slice of a table read from index files (see fga_dgix_build_range)
extern "C" int fga_dgix_upload_range(fga_dev *dev, const fga_gix *X, int64_t pbeg, int64_t pend, fga_dgix **out)
{ if (pbeg < 0 || pend > FGA_NPREFIX || pbeg >= pend)
    { fga_set_error("fga_dgix_upload_range: bad prefix range");
      *out = NULL;
      return 1;
    }
  return dgix_upload_impl(dev,X,pbeg,pend,out);
}

extern "C" int64_t fga_dgix_nents(const fga_dgix *D) { return D == NULL ? 0 : D->nents; }

extern "C" void fga_dgix_free(fga_dgix *D)
{ if (D == NULL) return;
  hipSetDevice(D->dev->device);
  fga_dgix_free_views(D);
  hipFree(D->table);
  hipFree(D->index);
  free(D);
}

extern "C" int64_t fga_seeds_count(const fga_dseeds *S)    { return S->count; }
extern "C" int64_t fga_seeds_plen_sum(const fga_dseeds *S) { return S->tseed; }

extern "C" int fga_seeds_download(const fga_dseeds *S, fga_seed *host, int64_t max)
{ FGA_HIP(hipSetDevice(S->dev->device));
  if (S->valid == NULL)
    { int64_t n = S->count < S->capacity ? S->count : S->capacity;
      if (n > max) n = max;
      if (n > 0)
        FGA_HIP(hipMemcpy(host,S->seeds,sizeof(fga_seed)*(size_t) n,hipMemcpyDeviceToHost));
      return 0;
    }
  // a buffer with holes: block by block, the valid head of each
  const int64_t ext = fga_seeds_extent(S), nb = (ext + FGA_SEED_BLOCK - 1) / FGA_SEED_BLOCK;
  std::vector<uint16_t> v((size_t) nb + 1);
  std::vector<fga_seed> all((size_t) ext + 1);
  FGA_HIP(hipMemcpy(v.data(),S->valid,sizeof(uint16_t)*(size_t) nb,hipMemcpyDeviceToHost));
  if (ext > 0)
    FGA_HIP(hipMemcpy(all.data(),S->seeds,sizeof(fga_seed)*(size_t) ext,hipMemcpyDeviceToHost));
  int64_t o = 0;
  for (int64_t b = 0; b < nb && o < max; b++)
    { int64_t k = v[(size_t) b];
      if (b*FGA_SEED_BLOCK + k > ext) k = ext - b*FGA_SEED_BLOCK;
      if (o + k > max) k = max - o;
      if (k > 0)
        memcpy(host+o,all.data() + b*FGA_SEED_BLOCK,sizeof(fga_seed)*(size_t) k);
      o += k;
    }
  return 0;
}

extern "C" void fga_seeds_free(fga_dseeds *S)
{ if (S == NULL) return;
  hipSetDevice(S->dev->device);
  fga_dev_release(S->dev,S->slot,S->seeds);
  fga_dev_release(S->dev,SLOT_VALID,S->valid);
  hipFree(S->dcount);
  free(S);
}
