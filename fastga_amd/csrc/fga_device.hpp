// fga_device.hpp -- internal device-side types of libfastga_amd (HIP, gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <utility>
#include <string.h>

#include "fga_host.h"
#include "fastga_amd.h"

#define FGA_HIP(call)                                                                       \
  do { hipError_t _e = (call);                                                              \
       if (_e != hipSuccess)                                                                \
         { fga_set_error("%s failed at %s:%d: %s",#call,__FILE__,__LINE__,hipGetErrorString(_e)); \
           return 1;                                                                        \
         }                                                                                  \
     } while (0)

struct fga_dev
  { int          device;
    hipStream_t  stream;
    hipEvent_t   ev0, ev1;
    int          ncu;
    float        last_ms[8];     // per-stage kernel time of the most recent call (HIP events)
    void        *pinned;          // pinned host staging buffer (key download)
    size_t       pinned_bytes;
    // what the last extension launch really needed (cells of the trace-point pool per hit-box base, output trace bytes
    // per base): the next launch over similar inputs starts from there instead of finding out by a repeated launch
    double       ext_cells_per_base, ext_tbytes_per_base;
    double       chain_density;   // small units with hits per key of the last chain scan (0: none yet): picks its kernel
    size_t       hbm_low_water;   // smallest free device memory seen at the stage boundaries (fga_dev_note_memory)
    int          host_threads;    // threads the host tails of the device stages may use (fga_dev_set_host_threads; 0 = 1)
  };
void fga_dev_note_memory(fga_dev *dev);
// every entry point of the C-ABI that takes a device context starts here: the device becomes the calling thread's current one
// and the context's stream the thread's "current stream" -- the stream an allocation made by this thread belongs to, which is
// what a release of the allocation waits for (fga_device.hip)
hipError_t fga_dev_enter(const fga_dev *dev);
// hipMemset on the calling thread's current stream.  (hipMemset itself returns before the fill has happened and runs on the
// legacy default stream, which is NOT ordered with the contexts' non-blocking streams: with several contexts at work in one
// process the fill could land after the kernel that was launched behind it had written the buffer.)
hipError_t fga_memset_here(void *ptr, int value, size_t bytes);

// Device memory of a MiB and more is a piece of a region the process keeps (fga_device.hip: the pool); smaller requests go
// to hipMalloc.  fga_pool_free takes either kind.  Every allocation of the library goes through these two.
hipError_t fga_pool_malloc(void **out, size_t bytes);
hipError_t fga_pool_free(void *ptr);
template <class T> static inline hipError_t fga_dmalloc(T **out, size_t bytes) { return fga_pool_malloc((void **) out,bytes); }
size_t fga_dev_largest(fga_dev *dev, size_t reserve);          // largest allocation that needs no region given back

// a table as the seed merge reads it: one array per field (fga_view.hip)
struct fga_view
  { uint64_t *K;          // (12-mer prefix & 0xff) << 56 | 56-bit suffix
    uint8_t  *L;          // lcp byte (first entry of a panel clamped to <= 11); NULL in a forward view
    uint8_t  *M;          // soft-mask byte
    uint32_t *P;          // position inside the contig
    void     *C;          // contig | sign, cw bytes each
    uint32_t *idx;        // [2^24] inclusive cumulative entry count per 12-mer prefix: its LOW 32 bits
    int64_t   n;
    int       cw;         // 1, 2 or 4
    uint32_t  car[4];     // car[k] = the first prefix whose cumulative count reaches (k+1) * 2^32 (2^24: none): the high part of
                          //   an absolute count is the number of car[] <= p.  Differences inside a tile are plain u32 arithmetic
    uint64_t  gen;        // a number no other view of this process has (fga_view_alloc): what caches key on -- the pool hands
  };                      //   addresses out again

// the carries of a view by value (kernel arguments), and the absolute count of the prefixes <= p (p = -1: 0)
struct fga_car { uint32_t c[4]; };
static inline fga_car fga_view_car(const fga_view &V) { fga_car r; for (int k = 0; k < 4; k++) r.c[k] = V.car[k]; return r; }
#ifdef __HIPCC__
__device__ __forceinline__ int64_t fga_idx_abs(const uint32_t *idx, const fga_car &car, int64_t p)
{ if (p < 0) return 0;
  const uint32_t q = (uint32_t) p;
  const int64_t hi = (q >= car.c[0]) + (q >= car.c[1]) + (q >= car.c[2]) + (q >= car.c[3]);
  return (hi << 32) | (int64_t) idx[p];
}
#endif
static inline int64_t fga_idx_hi(const fga_view &V, int64_t p)      // the high part alone (host: the low word is on the device)
{ if (p < 0) return 0;
  int64_t hi = 0;
  for (int k = 0; k < 4; k++) hi += ((uint32_t) p >= V.car[k]);
  return hi << 32;
}
int fga_view_set_carries(fga_dev *dev, const int64_t *idx64_device, fga_view *V);   // V->car from the table's 64-bit index

// device-resident genome index: the prefix index and the field arrays of the table (the on-disk bytes themselves only
// while the view is being made)
struct fga_dgix
  { fga_dev  *dev;
    uint8_t  *table;      // nents*ebytes raw entries (+ 64 bytes slack); NULL once the view exists
    int64_t  *index;      // [2^24] inclusive cumulative counts
    int64_t   nents;
    int       ebytes, postbytes, contbytes, nctg;
    int       legacy_cutoff;   // > 0: read from the pre-v1.3 layout, no k-mer with more positions than this is in it
    fga_view  view;       // all entries
    fga_view  fview;      // forward-strand entries only (made on first use as table 1 of a pair comparison)
    // the range cuts of the last merge launch with this table as table 1 (fga_merge.hip): a session repeats the same
    // comparison, and the cuts depend on the two prefix indices, the prefix range and the launch geometry only
    struct { uint64_t gen1, gen2; int pbeg, pend, nranges, nbig; int64_t base, total; int64_t *cuts; } cutc;
  };
int  fga_dgix_make_view(fga_dev *dev, fga_dgix *D, int keep_table);
int  fga_view_alloc(fga_view *V, int64_t n, int cont, int want_l);       // the field arrays of n entries (+ read slack), defined
int  fga_dgix_make_forward(fga_dev *dev, fga_dgix *D);
void fga_dgix_free_views(fga_dgix *D);

// One adaptive seed, 16 bytes (device + host layout of fga_seed in fastga_amd.h)
//   apos, bpos : in-contig positions as stored in the index payloads
//   actg       : A contig (length-sorted index) << 8 | plen
//   bctg       : B contig | (B entry's own sign bit) << 30 | (C-stream flag) << 31
enum { SLOT_SEEDS = 0, SLOT_SORT0, SLOT_SORT1, SLOT_HIST, SLOT_TILES, SLOT_CELLS, SLOT_TRACE, SLOT_ALNS,
       SLOT_TBYTES, SLOT_MISC, SLOT_VALID, SLOT_STAGE, SLOT_COUNT };
#define SLOT_BORROWED (-2)      // fga_dseeds.slot: the seeds are a stretch of somebody else's buffer (fga_seeds_view): not released
void *fga_dev_acquire(fga_dev *dev, int slot, size_t bytes);   // a work buffer (the slot names its purpose); NULL on failure
void  fga_dev_release(fga_dev *dev, int slot, void *ptr);
void *fga_dev_pinned(fga_dev *dev, size_t bytes);              // host pinned staging, grow-only
int   fga_radix_sort_u128(fga_dev *dev, uint4 *buf0, uint4 *buf1, int64_t n, int lowbit, int nbits, uint4 **sorted);
// exclusive prefix of n 32-bit counts into 64-bit offsets on dev->stream (fga_chain.hip: tile sums, their scan, the tiles
// again); tsum: scratch of (n + 4095) / 4096 + 1 words.  out[n] is not written.
void  fga_scan_counts(fga_dev *dev, const int32_t *cnt, int64_t n, int64_t *tsum, int64_t *out);

struct fga_dseeds
  { fga_dev  *dev;
    fga_seed *seeds;      // device buffer
    int64_t   capacity;
    int64_t   phys_capacity;   // slots actually allocated: capacity + room for the wave kernel's unused chunk tails
    int       slot;       // workspace slot the seed buffer came from (-1: own allocation)
    int64_t   tseed;      // sum of plen over all seeds
    int64_t   count;      // seeds produced (may exceed capacity -> overflow, buffer holds `capacity`)
    int64_t  *dcount;     // device counters: [0] slots handed out, [1] sum of plen, [2] slots left unused (holes)
    // The range-walking merge kernel hands out the buffer in 1024-seed blocks and never closes the unused tail of a
    // wavefront's last block: valid[b] = seeds in block b (1024 unless it is such a tail); every consumer -- the sort's
    // first pass, the routing kernels, the download -- skips the rest.  NULL: the buffer is dense.
    uint16_t *valid;
    int64_t   phys_count; // slots in use including the holes (== count for a dense buffer)
  };
#define FGA_SEED_BLOCK 1024
// slots a consumer has to look at
static inline int64_t fga_seeds_extent(const fga_dseeds *S)
{ const int64_t n = S->valid != NULL ? S->phys_count : S->count;
  return n < S->phys_capacity ? n : S->phys_capacity;
}

// sorted 128-bit diagonal records (fga_sort.hip)
struct fga_dkeys
  { fga_dev  *dev;
    uint4    *keys;        // x,y = low 64 bits; z,w = high 64 bits
    int64_t   count;
    size_t    alloc_bytes;
    int       slot;
    int       wa, wb, wd, wt;
    int64_t   amxpos, bmxpos;
  };

// device-resident genome: the .bps image (padded) and, for genome 1, its per-contig reverse complement
#define FGA_IMG_PAD 4096    // bytes of zero padding before and after a genome image (>= one LDS window)

struct fga_dgenome
  { fga_dev  *dev;
    uint8_t  *img, *img_rc;
    int64_t  *boff, *clen;
    int64_t  *hclen;      // host copy of the contig lengths (argument checks)
    int      *perm;
    int       nctg, nperm;
    int64_t   pad, maxctg;
  };

// stable order of n 64-bit keys (host): ord[k] = index of the k-th smallest key; LSD radix, 11 bits per pass, passes
// whose digit is constant are skipped.  Used where a comparison sort of ~10^6 units showed up in the stage times.
static inline void fga_radix_order(const uint64_t *keys, int64_t n, std::vector<int64_t> &ord)
{ std::vector<uint64_t> ka(keys,keys+n), kb((size_t) n);
  std::vector<int64_t>  va((size_t) n), vb((size_t) n);
  uint64_t all_or = 0, all_and = ~0ull;
  for (int64_t i = 0; i < n; i++)
    { va[(size_t) i] = i; all_or |= ka[(size_t) i]; all_and &= ka[(size_t) i]; }
  const uint64_t varying = all_or & ~all_and;
  for (int shift = 0; shift < 64; shift += 11)
    { if (((varying >> shift) & 0x7ff) == 0)
        continue;
      int64_t cnt[2049];
      memset(cnt,0,sizeof(cnt));
      for (int64_t i = 0; i < n; i++)
        cnt[((ka[(size_t) i] >> shift) & 0x7ff) + 1] += 1;
      for (int d = 0; d < 2048; d++)
        cnt[d+1] += cnt[d];
      for (int64_t i = 0; i < n; i++)
        { const int64_t d = cnt[(ka[(size_t) i] >> shift) & 0x7ff]++;
          kb[(size_t) d] = ka[(size_t) i]; vb[(size_t) d] = va[(size_t) i];
        }
      ka.swap(kb); va.swap(vb);
    }
  ord.swap(va);
}

