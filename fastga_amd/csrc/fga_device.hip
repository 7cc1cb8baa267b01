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
  (void) fga_dev_enter(d);
  *out = d;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// device memory: a pool of regions
// ---------------------------------------------------------------------------------------------------
// hipMalloc of tens of GB is not free: 10-28 ms per GB on most boxes and processes (a 54 GB buffer: 1.0-1.5 s; 64 GB after
// the hipFree of another buffer: 3.2 s; its hipFree: 1.7 s), 0.4 ms on others -- which one a process gets could not be
// predicted.  A 3 Gbp comparison allocates ~330 GB over its life (index staging, on-disk
// bytes, views, seeds, sort buffers, trace-point pool) and holds at most ~250 at a time, so every byte that is REUSED
// instead of freed and allocated again saves that time.  All device memory of a MiB and more therefore comes from regions
// this process keeps: a request takes the smallest free piece that holds it (split, the rest stays free), a release merges
// the piece with its free neighbours; a region goes back to the driver only when a new one cannot be had otherwise
// (fga_dev_trim: all regions that are entirely free), or when its device context closes.  Pieces are 2 MiB aligned.
// Like hipFree, releasing a piece waits for the device: the piece may be handed out again at once, to a copy on another
// stream.
#include <mutex>
#include "fga_pool.hpp"
#define POOL_MIN    ((size_t) 1 << 20)
#define POOL_ALIGN  ((size_t) 2 << 20)
#define POOL_MAXDEV 16
struct pool_state
  { fga_pool_core core;                       // pieces and regions (fga_pool.hpp: also run on the host by the tests)
    std::mutex mu;
  };
static pool_state g_pool[POOL_MAXDEV];

// seconds this process has spent inside hipMalloc / hipFree for the pool's regions: 0.3 ms per call, unless the driver has
// to finish clearing memory another process released before it can hand it out (fga_dev_driver_seconds)
static double g_driver_seconds = 0.;
static std::mutex g_driver_mu;

static int pool_hip_alloc(void **out, size_t bytes)
{ const double t0 = fga_wall();
  const hipError_t e = hipMalloc(out,bytes);
  { std::lock_guard<std::mutex> lk(g_driver_mu);
    g_driver_seconds += fga_wall() - t0;
  }
  if (e != hipSuccess)
    { (void) hipGetLastError();
      *out = NULL;
      return 1;
    }
  if (bytes >= ((size_t) 1 << 30))
    { char what[64];
      snprintf(what,sizeof(what),"hipMalloc %.1f GB",bytes*1e-9);
      fga_note(what,t0);
    }
  return 0;
}

static void pool_hip_release(void *ptr)
{ const double t0 = fga_wall();
  hipFree(ptr);
  { std::lock_guard<std::mutex> lk(g_driver_mu);
    g_driver_seconds += fga_wall() - t0;
  }
  fga_note("hipFree of an idle region",t0);
}

static const fga_pool_backend pool_hip = { pool_hip_alloc, pool_hip_release };

static pool_state *pool_here(void)
{ int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= POOL_MAXDEV) return NULL;
  return &g_pool[d];
}

// Who waits for what at a release.  hipFree waits for the whole device, and so did this pool's release -- which made
// sessions on different host threads take turns: every buffer one of them gave back waited for the kernels of all the others
// (eight comparisons in flight on one GPU ran 1.5 x faster than one after the other, not 6 x).  The library's rule is one HIP
// stream per device context, and a buffer is used by the stream of the context it was allocated under (where two contexts
// share one -- the seed exchange of fga_run_multi -- the reader synchronises its own stream before the owner is told to
// release).  So an allocation remembers the calling thread's current stream (fga_dev_enter), and its release waits for THAT
// stream (and the legacy default stream, which the few hipMemset calls at allocation time use).  Small requests (< 1 MiB) do
// not go back to hipFree either: they are kept in per-device lists by size class and handed out again.
static thread_local hipStream_t g_tls_stream = NULL;
static thread_local bool        g_tls_have = false;

hipError_t fga_dev_enter(const fga_dev *dev)
{ g_tls_stream = dev->stream; g_tls_have = true;
  return hipSetDevice(dev->device);
}

hipError_t fga_memset_here(void *ptr, int value, size_t bytes)
{ return hipMemsetAsync(ptr,value,bytes,g_tls_have ? g_tls_stream : (hipStream_t) NULL); }

#include <unordered_map>
struct pool_owner { hipStream_t stream; bool have; int klass; };       // klass >= 0: a small buffer of 256 << klass bytes
#define SMALL_CLASSES 13                                                // 256 B .. 1 MiB
#include <unordered_set>
struct small_state
  { std::unordered_map<void *,pool_owner> owner;                        // every live allocation of the device
    std::vector<void *> idle[SMALL_CLASSES];
    std::unordered_set<void *> parked;                                  // the small buffers in the idle lists: a SECOND release of one of
  };                                                                    //   them is ignored (it must never reach hipFree: the list still holds it)
static small_state g_small[POOL_MAXDEV];

static void wait_for_owner(const pool_owner &o, int owner_dev)
{ int cur = -1;
  const bool hop = owner_dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != owner_dev && hipSetDevice(owner_dev) == hipSuccess;
  static const int whole = getenv("FGA_POOL_DEVICE_SYNC") != NULL && atoi(getenv("FGA_POOL_DEVICE_SYNC")) != 0;
  if (o.have && !whole)
    { (void) hipStreamSynchronize(o.stream);
      (void) hipStreamSynchronize(NULL);
    }
  else
    (void) hipDeviceSynchronize();
  if (hop)
    (void) hipSetDevice(cur);
}

hipError_t fga_pool_malloc(void **out, size_t bytes)
{ *out = NULL;
  if (bytes == 0) bytes = 16;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= POOL_MAXDEV)
    return hipMalloc(out,bytes);
  pool_state *P = &g_pool[d];
  small_state *Q = &g_small[d];
  pool_owner o; o.stream = g_tls_stream; o.have = g_tls_have; o.klass = -1;
  if (bytes < POOL_MIN)
    { int k = 0;
      while (((size_t) 256 << k) < bytes) k += 1;
      o.klass = k;
      { std::lock_guard<std::mutex> lock(P->mu);
        if (!Q->idle[k].empty())
          { *out = Q->idle[k].back();
            Q->idle[k].pop_back();
            Q->parked.erase(*out);
            Q->owner[*out] = o;
            return hipSuccess;
          }
      }
      const hipError_t e = hipMalloc(out,(size_t) 256 << k);
      if (e != hipSuccess) { *out = NULL; return e; }
      std::lock_guard<std::mutex> lock(P->mu);
      Q->owner[*out] = o;
      return hipSuccess;
    }
  const size_t need = (bytes + POOL_ALIGN-1) / POOL_ALIGN * POOL_ALIGN;
  std::lock_guard<std::mutex> lock(P->mu);
  bool fresh;
  *out = P->core.take(need,pool_hip,&fresh);
  if (*out != NULL)
    Q->owner[*out] = o;
  return *out != NULL ? hipSuccess : hipErrorOutOfMemory;
}

hipError_t fga_pool_free(void *ptr)
{ if (ptr == NULL) return hipSuccess;
  // the pool of the calling thread's device first, then the others (a pointer belongs to at most one)
  int here = -1;
  if (hipGetDevice(&here) != hipSuccess || here < 0 || here >= POOL_MAXDEV) here = -1;
  for (int t = -1; t < POOL_MAXDEV; t++)
    { const int d = t < 0 ? here : t;
      if (d < 0 || (t >= 0 && d == here))
        continue;
      pool_state *P = &g_pool[d];
      small_state *Q = &g_small[d];
      std::unique_lock<std::mutex> lock(P->mu);
      auto it = Q->owner.find(ptr);
      if (it == Q->owner.end())
        continue;
      const pool_owner o = it->second;
      Q->owner.erase(it);                       // (a second release of the same pointer finds nothing and falls through)
      lock.unlock();
      // nothing in flight refers to the buffer any more: what hipFree guarantees, for the stream that used it
      wait_for_owner(o,d == here ? -1 : d);
      lock.lock();
      if (o.klass >= 0)
        { Q->idle[o.klass].push_back(ptr); Q->parked.insert(ptr); }
      else
        P->core.give(ptr);
      return hipSuccess;
    }
  // released before?  A small buffer that sits in an idle list is not the driver's to free
  for (int d = 0; d < POOL_MAXDEV; d++)
    { std::lock_guard<std::mutex> lock(g_pool[d].mu);
      if (g_small[d].parked.count(ptr) != 0)
        return hipSuccess;
    }
  // not an allocation of this library: the driver's to judge, and its verdict stays out of the sticky error state that the
  // stages' hipGetLastError() checks read
  const hipError_t e = hipFree(ptr);
  if (e != hipSuccess)
    (void) hipGetLastError();
  return e;
}

// the small buffers nobody uses go back to the driver (the calling thread's device)
static void small_trim(int d)
{ if (d < 0 || d >= POOL_MAXDEV) return;
  std::vector<void *> gone;
  { std::lock_guard<std::mutex> lock(g_pool[d].mu);
    for (int k = 0; k < SMALL_CLASSES; k++)
      { gone.insert(gone.end(),g_small[d].idle[k].begin(),g_small[d].idle[k].end());
        g_small[d].idle[k].clear();
      }
    g_small[d].parked.clear();
  }
  for (void *p : gone)
    hipFree(p);
}

// bytes in free pieces, and the largest of them
static void pool_idle(size_t *total, size_t *largest)
{ *total = *largest = 0;
  pool_state *P = pool_here();
  if (P == NULL) return;
  std::lock_guard<std::mutex> lock(P->mu);
  P->core.idle(total,largest);
}

// the regions nobody uses go back to the device (another library in the process -- torch's exchange buffers -- may need them)
extern "C" void fga_dev_trim(fga_dev *dev)
{ if (fga_dev_enter(dev) != hipSuccess) return;
  pool_state *P = pool_here();
  if (P == NULL) return;
  small_trim(dev->device);
  std::lock_guard<std::mutex> lock(P->mu);
  P->core.trim(pool_hip);
}

// device memory an allocation could get right now: free memory + what the pool's free pieces hold
extern "C" size_t fga_dev_available(fga_dev *dev)
{ size_t fr = 0, tot = 0, idle = 0, big = 0;
  if (fga_dev_enter(dev) != hipSuccess || hipMemGetInfo(&fr,&tot) != hipSuccess)
    return 0;
  pool_idle(&idle,&big);
  return fr + idle;
}

// the largest single allocation that can succeed without giving regions back: a free piece, or fresh memory less a reserve
size_t fga_dev_largest(fga_dev *dev, size_t reserve)
{ size_t fr = 0, tot = 0, idle = 0, big = 0;
  if (fga_dev_enter(dev) != hipSuccess || hipMemGetInfo(&fr,&tot) != hipSuccess)
    return 0;
  pool_idle(&idle,&big);
  fr = fr > reserve ? fr - reserve : 0;
  return fr > big ? fr : big;
}

// FGA_TIMING=1: wall-clock notes on stderr (allocations of a GB and more, the steps of an index build): where a
// human-scale run spends its host time
extern "C" void fga_note(const char *what, double since)
{ static int on = -1;
  if (on < 0) on = getenv("FGA_TIMING") != NULL && atoi(getenv("FGA_TIMING")) != 0;
  if (on)
    fprintf(stderr,"[fga timing] %-40s %9.1f ms\n",what,1e3*(fga_wall() - since));
}

// the pipeline's work buffers: `slot` says what the buffer is for (a name in the callers, nothing more: every buffer is a
// piece of the pool)
void *fga_dev_acquire(fga_dev *dev, int slot, size_t bytes)
{ (void) dev; (void) slot;
  void *p = NULL;
  if (fga_pool_malloc(&p,bytes > 0 ? bytes : 16) != hipSuccess)
    return NULL;
  return p;
}

void fga_dev_release(fga_dev *dev, int slot, void *ptr)
{ (void) dev; (void) slot;
  fga_pool_free(ptr);
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
{ if (fga_dev_enter(dev) != hipSuccess) return NULL;
  void *p = fga_dev_acquire(dev,SLOT_STAGE,bytes);
  if (p == NULL)
    fga_set_error("device allocation of %zu bytes (seed staging) failed",bytes);
  return p;
}

extern "C" void fga_dev_stage_release(fga_dev *dev, void *ptr)
{ fga_dev_release(dev,SLOT_STAGE,ptr); }

extern "C" int fga_dev_malloc(fga_dev *dev, size_t bytes, void **out)
{ *out = NULL;
  FGA_HIP(fga_dev_enter(dev));
  if (fga_pool_malloc(out,bytes > 0 ? bytes : 16) != hipSuccess)
    *out = NULL;
  if (*out == NULL)
    { fga_set_error("device allocation of %zu bytes failed: out of memory",bytes);
      return 1;
    }
  return 0;
}

extern "C" void fga_dev_free(fga_dev *dev, void *ptr)
{ if (ptr == NULL) return;
  fga_dev_enter(dev);
  fga_pool_free(ptr);
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

extern "C" double fga_dev_driver_seconds(void)
{ std::lock_guard<std::mutex> lk(g_driver_mu);
  return g_driver_seconds;
}

extern "C" int64_t fga_dev_peak_bytes(fga_dev *dev)
{ size_t fr = 0, tot = 0;
  if (fga_dev_enter(dev) != hipSuccess || hipMemGetInfo(&fr,&tot) != hipSuccess)
    return -1;
  fga_dev_note_memory(dev);
  return (int64_t) (tot - dev->hbm_low_water);
}

extern "C" int fga_dev_download(fga_dev *dev, void *host_dst, const void *device_src, size_t bytes)
{ FGA_HIP(fga_dev_enter(dev));
  if (bytes > 0)
    FGA_HIP(hipMemcpy(host_dst,device_src,bytes,hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int fga_dev_upload(fga_dev *dev, void *device_dst, const void *host_src, size_t bytes)
{ FGA_HIP(fga_dev_enter(dev));
  if (bytes > 0)
    FGA_HIP(hipMemcpy(device_dst,host_src,bytes,hipMemcpyHostToDevice));
  return 0;
}

extern "C" void fga_dev_close(fga_dev *d)
{ if (d == NULL) return;
  fga_dev_enter(d);
  hipStreamSynchronize(d->stream);
  fga_dev_trim(d);
  if (d->pinned != NULL) hipHostFree(d->pinned);
  hipEventDestroy(d->ev0);
  hipEventDestroy(d->ev1);
  // buffers that outlive their context no longer have a stream to wait for: their release waits for the device
  if (d->device >= 0 && d->device < POOL_MAXDEV)
    { std::lock_guard<std::mutex> lock(g_pool[d->device].mu);
      for (auto &kv : g_small[d->device].owner)
        if (kv.second.have && kv.second.stream == d->stream)
          kv.second.have = false;
    }
  if (g_tls_have && g_tls_stream == d->stream)
    { g_tls_have = false; g_tls_stream = NULL; }
  hipStreamDestroy(d->stream);
  free(d);
}

// direct loads / copies between this device and `peer` over xGMI (hipMemcpyPeerAsync works without, through the host); an
// access that is enabled already, or a peer that is this very device, is fine
extern "C" int fga_dev_enable_peer(fga_dev *d, int peer)
{ FGA_HIP(fga_dev_enter(d));
  if (peer == d->device)
    return 0;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can,d->device,peer) != hipSuccess || !can)
    { (void) hipGetLastError();
      return 0;                       // no direct path: the copies are staged by the runtime
    }
  const hipError_t e = hipDeviceEnablePeerAccess(peer,0);
  if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
    { fga_set_error("hipDeviceEnablePeerAccess(%d -> %d): %s",d->device,peer,hipGetErrorString(e));
      return 1;
    }
  (void) hipGetLastError();
  return 0;
}

extern "C" int fga_dev_device_count(void)
{ int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    { (void) hipGetLastError();
      return 0;
    }
  return n;
}

// the calling thread's current HIP device (-1: none can be told), and back: what an entry point that visits several devices
// leaves as it found it
extern "C" int fga_dev_current_device(void)
{ int d = -1;
  if (hipGetDevice(&d) != hipSuccess)
    { (void) hipGetLastError();
      return -1;
    }
  return d;
}

extern "C" void fga_dev_restore_device(int device)
{ if (device >= 0 && hipSetDevice(device) != hipSuccess)
    (void) hipGetLastError();
}

extern "C" int fga_dev_sync(fga_dev *d)
{ FGA_HIP(fga_dev_enter(d));
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
  FGA_HIP(fga_dev_enter(dev));
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
  if ((e = fga_dmalloc(&D->table,tbytes + 64)) != hipSuccess ||
      (e = fga_dmalloc(&D->index,sizeof(int64_t)*FGA_NPREFIX)) != hipSuccess)
    { fga_set_error("fga_dgix_upload: device allocation of %zu bytes failed: %s",tbytes,hipGetErrorString(e));
      fga_pool_free(D->table); fga_pool_free(D->index); free(D); free(sub);
      return 1;
    }
  if ((e = hipMemcpy(D->table,X->table + (size_t) lo*X->ebytes,tbytes,hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemsetAsync(D->table + tbytes,0,64,dev->stream)) != hipSuccess ||
      (e = hipMemcpy(D->index,hidx,sizeof(int64_t)*FGA_NPREFIX,hipMemcpyHostToDevice)) != hipSuccess)
    { fga_set_error("fga_dgix_upload: copy failed: %s",hipGetErrorString(e));
      fga_pool_free(D->table); fga_pool_free(D->index); free(D); free(sub);
      return 1;
    }
  free(sub);
  D->legacy_cutoff = X->legacy ? X->freq : 0;
  if (fga_dgix_make_view(dev,D,0))
    { fga_pool_free(D->table); fga_pool_free(D->index); free(D);
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
  fga_dev_enter(D->dev);
  fga_dgix_free_views(D);
  fga_pool_free(D->table);
  fga_pool_free(D->index);
  free(D);
}

extern "C" int64_t fga_seeds_count(const fga_dseeds *S)    { return S->count; }
extern "C" int64_t fga_seeds_plen_sum(const fga_dseeds *S) { return S->tseed; }

extern "C" int fga_seeds_download(const fga_dseeds *S, fga_seed *host, int64_t max)
{ FGA_HIP(fga_dev_enter(S->dev));
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
  fga_dev_enter(S->dev);
  if (S->slot != SLOT_BORROWED)
    fga_dev_release(S->dev,S->slot,S->seeds);
  fga_dev_release(S->dev,SLOT_VALID,S->valid);
  fga_pool_free(S->dcount);
  free(S);
}
