// fga_sort.hip -- seed -> diagonal record transform and device radix sort (gfx950).
//
// Replaces reimport_thread (FastGA.c:2641-2747) and rmsd_sort (RSDsort.c:292-377) of the reference.
//   reference record : {u8 lcp; u8 diag&63; anti[DBYTE]; diag>>6[DBYTE]; jcont[JCONT]}, one panel per A contig,
//                      sorted ascending reading each record from its LAST byte (RSDsort.c), i.e. by
//                      (jcont, diag>>6, anti, diag&63, lcp) -- separately for the N and the C stream.
//   here             : ONE 128-bit key per seed with the same fields packed most-significant first
//                      [strand | A contig | B contig | diag>>6 | anti | diag&63 | lcp]
//                      and one global LSD radix sort over the significant bits only.  Ascending key order
//                      restricted to one (strand, A contig) is exactly the reference's panel order.
//   N stream: diag = BMXPOS + (i-j), anti = i+j;  C stream: diag = MAXDAG - (i+j), anti = AMXPOS - (i-j)
//   (FastGA.c:2705-2712).
//
// LSD radix sort, 8-bit digits, three kernels per pass (tile histogram, scan, stable scatter).  The scatter ranks
// keys inside a wavefront with ballot-based digit matching (no per-thread histograms) and keeps per-wave digit
// counters in LDS; the first pass reads the 16-byte seeds and forms the key on the fly, so the transform
// costs no extra trip through HBM.  HBM-bound: (2 reads + 1 write) x 16 B x passes.

#include "fga_device.hpp"

#define ST        256            // threads
#define SITEMS    16             // keys per thread
#define STILE     (ST*SITEMS)    // keys per tile
#define SWAVES    (ST/64)

struct key_layout
  { int wa, wb, wd, wt;          // bit widths of A contig, B contig, diag bucket, anti
    int64_t amx, bmx;            // AMXPOS, BMXPOS
  };

struct u128 { uint64_t lo, hi; };

__device__ __forceinline__ u128 make_key(const fga_seed &s, const key_layout &L)
{ const int64_t i = s.apos, j = s.bpos;
  const uint32_t comp = s.bctg >> 31;
  const uint64_t actg = s.actg >> 8, lcp = s.actg & 0xff, bctg = s.bctg & 0x3fffffffu;
  int64_t diag, anti;
  if (comp)
    { diag = (L.amx + L.bmx) - (i + j);
      anti = L.amx - (i - j);
    }
  else
    { diag = L.bmx + (i - j);
      anti = i + j;
    }
  // pack LSB first: lcp(6) drem(6) anti(wt) bucket(wd) bctg(wb) actg(wa) strand(1)
  unsigned __int128 k = 0;
  int sh = 0;
  k |= (unsigned __int128) (lcp & 63);                     sh += 6;
  k |= (unsigned __int128) (uint64_t) (diag & 63) << sh;   sh += 6;
  k |= (unsigned __int128) (uint64_t) anti << sh;          sh += L.wt;
  k |= (unsigned __int128) (uint64_t) (diag >> 6) << sh;   sh += L.wd;
  k |= (unsigned __int128) bctg << sh;                     sh += L.wb;
  k |= (unsigned __int128) actg << sh;                     sh += L.wa;
  k |= (unsigned __int128) comp << sh;
  u128 r;
  r.lo = (uint64_t) k;
  r.hi = (uint64_t) (k >> 64);
  return r;
}

__device__ __forceinline__ uint32_t digit_of(const u128 &k, int shift)
{ if (shift >= 64)
    return (uint32_t) (k.hi >> (shift-64)) & 0xff;
  uint64_t v = k.lo >> shift;
  if (shift > 56)
    v |= k.hi << (64-shift);
  return (uint32_t) v & 0xff;
}

template <bool FROM_SEEDS>
__device__ __forceinline__ u128 load_key(const void *in, int64_t i, const key_layout &L)
{ if (FROM_SEEDS)
    return make_key(((const fga_seed *) in)[i],L);
  const uint4 v = ((const uint4 *) in)[i];
  u128 r;
  r.lo = ((uint64_t) v.y << 32) | v.x;
  r.hi = ((uint64_t) v.w << 32) | v.z;
  return r;
}

// (1) per-tile digit histogram -> hist[digit*ntiles + tile]
// `valid` (first pass over a seed buffer with holes only): seeds in each 1024-slot block, the rest of the block is skipped
template <bool FROM_SEEDS>
__global__ __launch_bounds__(ST)
void sort_hist_kernel(const void *in, int64_t n, int shift, key_layout L, uint32_t *hist, int ntiles, const uint16_t *valid)
{ __shared__ uint32_t h[256];
  const int tile = blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t) tile * STILE;
  #pragma unroll 4
  for (int r = 0; r < SITEMS; r++)
    { int64_t i = base + r*ST + threadIdx.x;
      if (i < n && (!FROM_SEEDS || valid == NULL || (int) (i & 1023) < (int) valid[i >> 10]))
        { u128 k = load_key<FROM_SEEDS>(in,i,L);
          atomicAdd(&h[digit_of(k,shift)],1u);
        }
    }
  __syncthreads();
  hist[(int64_t) threadIdx.x * ntiles + tile] = h[threadIdx.x];
}

// (2) exclusive scan of hist (256*ntiles values): chunk-local scan + chunk totals, scan of totals, and the
//     scatter adds the chunk offset itself.
#define SCAN_T   1024
#define SCAN_PER 16
#define SCAN_CH  (SCAN_T*SCAN_PER)

__global__ __launch_bounds__(SCAN_T)
void sort_scan_local_kernel(uint32_t *hist, int64_t m, uint32_t *sums)
{ __shared__ uint32_t wsum[SCAN_T/64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = (int64_t) blockIdx.x * SCAN_CH + (int64_t) tid * SCAN_PER;
  uint32_t v[SCAN_PER], s = 0;
  #pragma unroll
  for (int q = 0; q < SCAN_PER; q++)
    { int64_t i = base + q;
      v[q] = (i < m) ? hist[i] : 0;
      s += v[q];
    }
  uint32_t inc = s;
  #pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    { uint32_t y = __shfl_up(inc,d,64);
      if (lane >= d) inc += y;
    }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t off = 0, tot = 0;
  #pragma unroll
  for (int w = 0; w < SCAN_T/64; w++)
    { uint32_t t = wsum[w];
      if (w < wave) off += t;
      tot += t;
    }
  uint32_t run = off + inc - s;
  #pragma unroll
  for (int q = 0; q < SCAN_PER; q++)
    { int64_t i = base + q;
      if (i < m) hist[i] = run;
      run += v[q];
    }
  if (tid == 0)
    sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_T)
void sort_scan_sums_kernel(uint32_t *sums, int nch)        // exclusive scan, single workgroup
{ __shared__ uint32_t wsum[SCAN_T/64];
  __shared__ uint32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nch; base += SCAN_T)
    { int i = base + tid;
      uint32_t v = (i < nch) ? sums[i] : 0, inc = v;
      #pragma unroll
      for (int d = 1; d < 64; d <<= 1)
        { uint32_t y = __shfl_up(inc,d,64);
          if (lane >= d) inc += y;
        }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      uint32_t off = carry;
      for (int w = 0; w < wave; w++)
        off += wsum[w];
      if (i < nch) sums[i] = off + inc - v;
      __syncthreads();
      if (tid == SCAN_T-1)
        carry = off + inc;
      __syncthreads();
    }
}

// (3) stable scatter
template <bool FROM_SEEDS>
__global__ __launch_bounds__(ST)
void sort_scatter_kernel(const void *in, uint4 *out, int64_t n, int shift, key_layout L,
                         const uint32_t *hist, const uint32_t *sums, int ntiles, const uint16_t *valid)
{ __shared__ uint32_t wcnt[SWAVES][256];      // per-wave digit counts, then per-wave digit bases
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int x = tid; x < SWAVES*256; x += ST)
    (&wcnt[0][0])[x] = 0;
  __syncthreads();

  // item order inside the tile: wave-major, then round, then lane  (coalesced 64-key loads per round)
  const int64_t wbase = (int64_t) tile * STILE + (int64_t) wave * (64*SITEMS);
  // a wavefront's 1024 items are exactly one block of the seed buffer
  static_assert(64*SITEMS == 1024,"one seed block per wavefront");
  const int vcount = (FROM_SEEDS && valid != NULL && wbase < n) ? (int) valid[wbase >> 10] : 64*SITEMS;
  u128     key[SITEMS];
  uint16_t rank[SITEMS];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64-lane));
  #pragma unroll
  for (int r = 0; r < SITEMS; r++)
    { int64_t i = wbase + r*64 + lane;
      bool ok = i < n && r*64 + lane < vcount;
      uint32_t d = 256;
      if (ok)
        { key[r] = load_key<FROM_SEEDS>(in,i,L);
          d = digit_of(key[r],shift);
        }
      // lanes with the same digit (invalid lanes form their own group and are ignored)
      uint64_t peers = __ballot(ok);
      #pragma unroll
      for (int b = 0; b < 8; b++)
        { uint64_t m = __ballot((d >> b) & 1);
          peers &= ((d >> b) & 1) ? m : ~m;
        }
      uint32_t before = __popcll(peers & lt);
      uint32_t basec = 0;
      if (ok)
        basec = wcnt[wave][d];
      rank[r] = (uint16_t) (basec + before);
      // the last peer publishes the new count (one writer per digit)
      if (ok && (peers >> lane) >> 1 == 0)
        wcnt[wave][d] = basec + before + 1;
    }
  __syncthreads();
  // per-wave exclusive bases per digit: thread d handles digit d
  { const int64_t hi = (int64_t) tid * ntiles + tile;
    uint32_t run = hist[hi] + sums[hi / SCAN_CH];
    #pragma unroll
    for (int w = 0; w < SWAVES; w++)
      { uint32_t c = wcnt[w][tid];
        wcnt[w][tid] = run;
        run += c;
      }
  }
  __syncthreads();
  #pragma unroll
  for (int r = 0; r < SITEMS; r++)
    { int64_t i = wbase + r*64 + lane;
      if (i < n && r*64 + lane < vcount)
        { uint32_t d = digit_of(key[r],shift);
          int64_t pos = (int64_t) wcnt[wave][d] + rank[r];
          uint4 v;
          v.x = (uint32_t) key[r].lo; v.y = (uint32_t) (key[r].lo >> 32);
          v.z = (uint32_t) key[r].hi; v.w = (uint32_t) (key[r].hi >> 32);
          out[pos] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// One-sweep passes (chained scan with decoupled look-back): a pass reads the keys once and writes them once.
//   os_first_hist   digit histogram of the first pass (the only extra sweep; with FROM_SEEDS it reads the 16-byte seeds)
//   os_scan256      exclusive scan of the 256 global digit counts -> where every digit's run starts
//   os_pass         a workgroup takes a tile by ticket (so every earlier tile is already running), ranks its keys like
//                   sort_scatter_kernel, publishes its 256 digit counts, looks back through the earlier tiles' entries
//                   until it meets an inclusive prefix, publishes its own inclusive prefix, scatters -- and counts the
//                   NEXT pass's digits of the keys it writes, so the next pass needs no histogram sweep
// A status entry is one 8-byte word {pass stamp : 8 | state : 2 | value : 54}, written and polled with relaxed
// agent-scope atomics (the granule hand-off of MI355X_MICROARCH.md: one sc1 store, sc1 polls; valid across XCDs).  The
// stamp makes entries of earlier passes read as "not there yet", so the array is cleared once, not per pass.
// Traffic per pass 2 n x 16 B instead of 3 n.
// ---------------------------------------------------------------------------------------------------
#define OS_LOCAL     1ull
#define OS_INCL      2ull
#define OS_PACK(stamp,state,val)  (((unsigned long long) (stamp) << 56) | ((unsigned long long) (state) << 54) | (unsigned long long) (val))
#define OS_STAMP(s)  ((int) ((s) >> 56))
#define OS_STATE(s)  (((s) >> 54) & 3ull)
#define OS_VALUE(s)  ((s) & ((1ull << 54) - 1))

template <bool FROM_SEEDS>
__global__ __launch_bounds__(ST)
void os_first_hist_kernel(const void *in, int64_t n, int shift, key_layout L, unsigned long long *ghist, const uint16_t *valid)
{ __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile*STILE < n; tile += gridDim.x)
    { const int64_t base = tile * STILE;
      #pragma unroll 4
      for (int r = 0; r < SITEMS; r++)
        { const int64_t i = base + r*ST + threadIdx.x;
          if (i < n && (!FROM_SEEDS || valid == NULL || (int) (i & 1023) < (int) valid[i >> 10]))
            { const u128 k = load_key<FROM_SEEDS>(in,i,L);
              atomicAdd(&h[digit_of(k,shift)],1u);
            }
        }
    }
  __syncthreads();
  if (h[threadIdx.x] != 0)
    atomicAdd(ghist + threadIdx.x,(unsigned long long) h[threadIdx.x]);
}

// ghist[256] counts -> gbase[256] exclusive starts; clears the counts of the pass after next and the ticket
__global__ __launch_bounds__(256)
void os_scan256_kernel(const unsigned long long *ghist, unsigned long long *gbase, unsigned long long *clear, unsigned int *ticket)
{ __shared__ unsigned long long t[256];
  t[threadIdx.x] = ghist[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0)
    { unsigned long long run = 0;
      for (int d = 0; d < 256; d++) { const unsigned long long v = t[d]; t[d] = run; run += v; }
    }
  if (threadIdx.x < 8)
    ticket[threadIdx.x*32] = 0;                 // the eight ticket queues of a pass (OS_TICKET_STRIDE), queue 0 = the only one of os_pass_kernel
  __syncthreads();
  gbase[threadIdx.x] = t[threadIdx.x];
  if (clear != NULL)
    clear[threadIdx.x] = 0;
}

#define OS_ITEMS   SITEMS             // a one-sweep tile is a tile of the three-kernel passes: 4096 keys
#define OS_TILE    STILE
// (Tried and dropped: 2048-key tiles whose keys are first put into digit order in LDS, so that the write-out is runs of
//  consecutive slots -- 6.1 ms instead of 4.4 ms on the bench pair, 152 instead of 121 ms on 0.55 G keys: the scattered
//  16-byte stores are not what bounds a pass.)

template <bool FROM_SEEDS>
__global__ __launch_bounds__(ST)
void os_pass_kernel(const void *in, uint4 *out, int64_t n, int shift, int next_shift, key_layout L, int stamp,
                    const unsigned long long *gbase, unsigned long long *status, unsigned long long *next_hist,
                    unsigned int *ticket, const uint16_t *valid)
{ __shared__ uint32_t wcnt[SWAVES][256];      // per-wave digit counts, then per-wave digit bases inside the tile
  __shared__ unsigned long long dbase[256];   // where the tile's keys of every digit start in `out`
  __shared__ uint32_t nh[256];                // digits of the next pass among the keys written
  __shared__ int tile_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0)
    tile_s = (int) atomicAdd(ticket,1u);
  for (int x = tid; x < SWAVES*256; x += ST)
    (&wcnt[0][0])[x] = 0;
  nh[tid] = 0;
  __syncthreads();
  const int tile = tile_s;

  const int64_t wbase = (int64_t) tile * OS_TILE + (int64_t) wave * (64*OS_ITEMS);
  static_assert(64*OS_ITEMS == 1024,"one seed block per wavefront");
  const int vcount = (FROM_SEEDS && valid != NULL && wbase < n) ? (int) valid[wbase >> 10] : 64*OS_ITEMS;
  u128     key[OS_ITEMS];
  uint16_t rank[OS_ITEMS];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64-lane));
  #pragma unroll
  for (int r = 0; r < OS_ITEMS; r++)
    { const int64_t i = wbase + r*64 + lane;
      const bool ok = i < n && r*64 + lane < vcount;
      uint32_t d = 256;
      if (ok)
        { key[r] = load_key<FROM_SEEDS>(in,i,L);
          d = digit_of(key[r],shift);
        }
      uint64_t peers = __ballot(ok);
      #pragma unroll
      for (int b = 0; b < 8; b++)
        { const uint64_t m = __ballot((d >> b) & 1);
          peers &= ((d >> b) & 1) ? m : ~m;
        }
      const uint32_t before = __popcll(peers & lt);
      uint32_t basec = 0;
      if (ok)
        basec = wcnt[wave][d];
      rank[r] = (uint16_t) (basec + before);
      if (ok && (peers >> lane) >> 1 == 0)
        wcnt[wave][d] = basec + before + 1;
    }
  __syncthreads();
  // thread d: the tile's count of digit d, published; per-wave bases; look-back over the earlier tiles
  { uint32_t run = 0;
    #pragma unroll
    for (int w = 0; w < SWAVES; w++)
      { const uint32_t c = wcnt[w][tid];
        wcnt[w][tid] = run;
        run += c;
      }
    unsigned long long *mine = status + (size_t) tile*256 + tid;
    if (tile > 0)
      __hip_atomic_store(mine,OS_PACK(stamp,OS_LOCAL,run),__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    // (Requesting the status entries of 4 / 8 / 16 predecessors together instead of one per step -- a step is a dependent
    //  agent-scope load -- changes nothing at 48.6 M keys and costs 2-20 % at 0.55 G: the walk is not what bounds a pass.)
    unsigned long long excl = 0;
    for (int t = tile-1; t >= 0; t--)
      { const unsigned long long *p = status + (size_t) t*256 + tid;
        unsigned long long sv = __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
        while (OS_STAMP(sv) != stamp || OS_STATE(sv) == 0)
          { __builtin_amdgcn_s_sleep(1);
            sv = __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
          }
        excl += OS_VALUE(sv);
        if (OS_STATE(sv) == OS_INCL)
          break;
      }
    __hip_atomic_store(mine,OS_PACK(stamp,OS_INCL,excl + run),__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
    dbase[tid] = gbase[tid] + excl;
  }
  __syncthreads();
  #pragma unroll
  for (int r = 0; r < OS_ITEMS; r++)
    { const int64_t i = wbase + r*64 + lane;
      if (i < n && r*64 + lane < vcount)
        { const uint32_t d = digit_of(key[r],shift);
          const int64_t pos = (int64_t) dbase[d] + wcnt[wave][d] + rank[r];
          uint4 v;
          v.x = (uint32_t) key[r].lo; v.y = (uint32_t) (key[r].lo >> 32);
          v.z = (uint32_t) key[r].hi; v.w = (uint32_t) (key[r].hi >> 32);
          out[pos] = v;
          if (next_hist != NULL)
            atomicAdd(&nh[digit_of(key[r],next_shift)],1u);
        }
    }
  if (next_hist != NULL)
    { __syncthreads();
      if (nh[tid] != 0)
        atomicAdd(next_hist + tid,(unsigned long long) nh[tid]);
    }
}

// ---------------------------------------------------------------------------------------------------
// One-sweep pass with wide tiles whose keys leave in tile order (round 4).  What bounds os_pass_kernel is not its look-back
// and not its instruction count but the stores (tools/ubench/sort_bench.hip, profiles/r04_sort_*): a 64-byte block written
// whole by 4 adjacent lanes of ONE store instruction goes at the speed of a copy (5.3 TB/s read + write); the same bytes written
// record by record, or in runs that begin and end inside a block, pay a read-modify-write per piece (2.4 / 2.7 TB/s).  A
// 4096-key tile holds 16 keys per digit and scatters them from registers: every store instruction touches 64 blocks.  Here
//   * a tile is NW x 1024 keys (NW wavefronts: 8192 / 16384 keys, runs of 32 / 64 keys per digit),
//   * the keys go through LDS in the order they have in the output, 2048 slots at a time, and leave it slot by slot:
//     consecutive lanes write consecutive records of a digit's run, so only the two ends of a run share their block.
// Ranking, status words and look-back are os_pass_kernel's (4 / 2 times fewer tiles to walk over).
// Measured (tools/ubench/sort_bench, profiles/r04_sort_wide_tiles.txt): 8 wavefronts 3.1 TB/s per pass (48.6 M keys 4.4 -> 3.5
// ms, 0.55 G keys 56 -> 39.8 ms); 16 wavefronts (one workgroup per CU) no better.  Tried on top and dropped: every run staged
// at its own phase modulo 4 (padding slots between the runs), so that a block of `out` is an ALIGNED group of 4 lanes --
// 3.7 / 38.8 ms: it is the partial blocks at the two ends of every run (2 of 9 at 32 keys per run) that still cost, not how
// the lanes of a full block are grouped.  What would remove them is carrying a run's last (position mod 4) keys over to the
// next tile that has keys of the digit (the block-aligned pass of this round's history, c447788): not rebuilt.
// ---------------------------------------------------------------------------------------------------
#define OSW_STAGE   2048

template <bool FROM_SEEDS, int NW, int KPT>
__global__ __launch_bounds__(NW*64) __attribute__((amdgpu_waves_per_eu(KPT <= 8 ? 8 : 4,KPT <= 8 ? 8 : 4)))
void osw_pass_kernel(const void *in, uint4 *out, int64_t n, int shift, int next_shift, key_layout L, int stamp,
                     const unsigned long long *gbase, unsigned long long *status, unsigned long long *next_hist,
                     unsigned int *ticket, const uint16_t *valid, unsigned chunk)
{ constexpr int NT = NW*64, WK = 64*KPT, TILE = NW*WK, SPT = OSW_STAGE/NT;      // KPT keys per thread, WK per wavefront
  __shared__ uint32_t wcnt[NW][256];          // per-wave digit counts, then per-wave digit bases inside the tile
  __shared__ long long off[256];              // position in `out` of the digit's first key of this tile, minus its tile position
  __shared__ uint32_t dstart[256];            // tile position of the digit's first key
  __shared__ uint32_t nh[256];                // digits of the next pass among the keys written
  __shared__ uint4 stage[OSW_STAGE];
  __shared__ uint32_t wsum[4];
  __shared__ int tile_s, total_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Tiles are taken from eight queues, one per XCD (workgroup b runs on XCD b mod 8): queue x hands out the tiles of the
  // chunks x, x+8, x+16, .. of `chunk` consecutive tiles, in order.  Neighbouring tiles then share an XCD's L2, where the
  // two partial blocks at the ends of their runs of a digit meet and leave as one whole block (32-key runs: 3.3 -> 4.2 TB/s
  // in tools/ubench/sort_bench's store patterns).  Forward progress does not depend on where or in which order workgroups
  // start, only on this: a queue's tickets map to increasing tiles, so the tiles handed out AHEAD of the smallest tile not
  // yet handed out are at most 7 x chunk (one chunk per other queue); they wait in their look-back, everything else that
  // is resident works on smaller tiles, finishes and makes room for the workgroup that takes that tile.  The host keeps
  // 7 x chunk well below the workgroups the device holds at once.  A queue that has run out sends the workgroup to the
  // next one (as many workgroups as tiles are launched).
  if (tid == 0)
    { int t = -1;
      for (int k = 0; k < 8 && t < 0; k++)
        { const unsigned q = (blockIdx.x + (unsigned) k) & 7u;
          const unsigned tk = atomicAdd(ticket + q*32,1u);
          const long long cand = ((long long) (tk / chunk)*8 + q)*chunk + tk % chunk;
          if (cand*TILE < n)
            t = (int) cand;
        }
      tile_s = t;
    }
  for (int x = tid; x < NW*256; x += NT)
    (&wcnt[0][0])[x] = 0;
  if (tid < 256) nh[tid] = 0;
  __syncthreads();
  const int tile = tile_s;
  if (tile < 0)                               // (cannot happen: tiles and workgroups are as many)
    return;

  // a wavefront's WK items lie inside one 1024-slot block of the seed buffer; boff = where in it
  const int64_t wbase = (int64_t) tile * TILE + (int64_t) wave * WK;
  const int boff = (int) (wbase & 1023);
  const int vcount = (FROM_SEEDS && valid != NULL && wbase < n) ? (int) valid[wbase >> 10] - boff : WK;
  u128     key[KPT];
  uint16_t rank[KPT];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64-lane));
  #pragma unroll
  for (int r = 0; r < KPT; r++)
    { const int64_t i = wbase + r*64 + lane;
      const bool ok = i < n && r*64 + lane < vcount;
      uint32_t d = 256;
      if (ok)
        { key[r] = load_key<FROM_SEEDS>(in,i,L);
          d = digit_of(key[r],shift);
        }
      uint64_t peers = __ballot(ok);
      #pragma unroll
      for (int b = 0; b < 8; b++)
        { const uint64_t m = __ballot((d >> b) & 1);
          peers &= ((d >> b) & 1) ? m : ~m;
        }
      const uint32_t before = __popcll(peers & lt);
      uint32_t basec = 0;
      if (ok)
        basec = wcnt[wave][d];
      rank[r] = (uint16_t) (basec + before);
      if (ok && (peers >> lane) >> 1 == 0)
        wcnt[wave][d] = basec + before + 1;
    }
  __syncthreads();
  // thread d < 256: the tile's count of digit d, published; per-wave bases; look-back; where the digit's run begins in the tile
  if (tid < 256)
    { uint32_t run = 0;
      #pragma unroll
      for (int w = 0; w < NW; w++)
        { const uint32_t c = wcnt[w][tid];
          wcnt[w][tid] = run;
          run += c;
        }
      unsigned long long *mine = status + (size_t) tile*256 + tid;
      if (tile > 0)
        __hip_atomic_store(mine,OS_PACK(stamp,OS_LOCAL,run),__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      uint32_t inc = run;                       // exclusive scan of the counts over the digits (waves 0..3)
      #pragma unroll
      for (int d = 1; d < 64; d <<= 1)
        { const uint32_t y = __shfl_up(inc,d,64);
          if (lane >= d) inc += y;
        }
      if (lane == 63) wsum[wave] = inc;
      unsigned long long excl = 0;
      for (int t = tile-1; t >= 0; t--)
        { const unsigned long long *p = status + (size_t) t*256 + tid;
          unsigned long long sv = __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
          while (OS_STAMP(sv) != stamp || OS_STATE(sv) == 0)
            { __builtin_amdgcn_s_sleep(1);
              sv = __hip_atomic_load(p,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
            }
          excl += OS_VALUE(sv);
          if (OS_STATE(sv) == OS_INCL)
            break;
        }
      __hip_atomic_store(mine,OS_PACK(stamp,OS_INCL,excl + run),__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);
      dstart[tid] = inc - run;                  // completed below with the earlier waves' totals
      off[tid] = (long long) (gbase[tid] + excl);
    }
  __syncthreads();
  if (tid < 256)
    { uint32_t o = 0;
      for (int w = 0; w < wave; w++) o += wsum[w];
      const uint32_t ds = dstart[tid] + o;
      dstart[tid] = ds;
      off[tid] -= (long long) ds;
      if (tid == 255)
        total_s = (int) (wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    }
  __syncthreads();
  // every key's position in the tile (over the rank: 16 bits each)
  #pragma unroll
  for (int r = 0; r < KPT; r++)
    { const int64_t i = wbase + r*64 + lane;
      if (i < n && r*64 + lane < vcount)
        { const uint32_t d = digit_of(key[r],shift);
          rank[r] = (uint16_t) (dstart[d] + wcnt[wave][d] + rank[r]);
        }
      else
        rank[r] = 0xffff;
    }
  const int total = total_s;
  for (int q = 0; q*OSW_STAGE < TILE; q++)
    { if (q*OSW_STAGE >= total)
        break;
      #pragma unroll
      for (int r = 0; r < KPT; r++)
        if ((rank[r] >> 11) == q && rank[r] != 0xffff)
          { uint4 v;
            v.x = (uint32_t) key[r].lo; v.y = (uint32_t) (key[r].lo >> 32);
            v.z = (uint32_t) key[r].hi; v.w = (uint32_t) (key[r].hi >> 32);
            stage[rank[r] & (OSW_STAGE-1)] = v;
          }
      __syncthreads();
      #pragma unroll
      for (int j = 0; j < SPT; j++)
        { const int sl = j*NT + tid, p = q*OSW_STAGE + sl;
          if (p < total)
            { const uint4 v = stage[sl];
              u128 k;
              k.lo = ((uint64_t) v.y << 32) | v.x; k.hi = ((uint64_t) v.w << 32) | v.z;
              out[off[digit_of(k,shift)] + p] = v;
              if (next_hist != NULL)
                atomicAdd(&nh[digit_of(k,next_shift)],1u);
            }
        }
      __syncthreads();
    }
  if (next_hist != NULL && tid < 256 && nh[tid] != 0)
    atomicAdd(next_hist + tid,(unsigned long long) nh[tid]);
}

// the passes of one sort, one-sweep; `first` reads seeds (FROM_SEEDS) or keys; buffers alternate.  Enqueued on the stream.
// work: 3 x 256 unsigned long long (two digit histograms, the digit bases), the tickets (eight queues, 128 bytes apart: the
// atomics of one cache line are served one after the other), then status (256 words per tile)
#define OS_TICKET_WORDS 128         // unsigned long long words reserved for the tickets
#define OS_TICKET_STRIDE 32         // unsigned int words between two queues' tickets
static size_t os_work_bytes(int64_t ntiles)
{ return sizeof(unsigned long long)*(3*256 + OS_TICKET_WORDS + 256*(size_t) ntiles); }

// FGA_SORT_WIDE = 0 | 8 | 16: wavefronts per tile of the one-sweep passes (0: the 4096-key tiles of os_pass_kernel); read once
static int os_wide()
{ static int v = -1;
  if (v < 0)
    { const char *e = getenv("FGA_SORT_WIDE");
      v = e != NULL ? atoi(e) : 8;
      if (v != 0 && v != 8 && v != 16) v = 8;
    }
  return v;
}
#define KPT16 8          // keys per thread of the 16-wavefront tiles: 8192-key tiles like the 8-wavefront ones, twice the wavefronts per CU
static int64_t os_tile_keys() { return os_wide() == 0 ? OS_TILE : (os_wide() == 16 ? (int64_t) 16*64*KPT16 : (int64_t) os_wide()*1024); }

static void os_sort(fga_dev *dev, const void *first, bool from_seeds, int64_t next, const uint16_t *valid, key_layout L,
                    uint4 *buf0, uint4 *buf1, int64_t n, int lowbit, int npass, void *work, int64_t ntiles_max, uint4 **sorted)
{ unsigned long long *hist[2] = { (unsigned long long *) work, (unsigned long long *) work + 256 };
  unsigned long long *gbase = (unsigned long long *) work + 512;
  unsigned int *ticket = (unsigned int *) ((unsigned long long *) work + 768);
  unsigned long long *status = (unsigned long long *) work + 768 + OS_TICKET_WORDS;
  hipMemsetAsync(work,0,os_work_bytes(ntiles_max),dev->stream);
  const void *src = first;
  uint4 *dst = buf0;
  int64_t cnt = from_seeds ? next : n;                     // slots the first pass looks at
  { int grid = (int) ((cnt + STILE - 1) / STILE);
    if (grid > dev->ncu*8) grid = dev->ncu*8;
    if (grid < 1) grid = 1;
    if (from_seeds)
      hipLaunchKernelGGL(os_first_hist_kernel<true>,dim3(grid),dim3(ST),0,dev->stream,src,cnt,lowbit,L,hist[0],valid);
    else
      hipLaunchKernelGGL(os_first_hist_kernel<false>,dim3(grid),dim3(ST),0,dev->stream,src,cnt,lowbit,L,hist[0],(const uint16_t *) NULL);
  }
  for (int p = 0; p < npass; p++)
    { const int shift = lowbit + 8*p;
      const int64_t m = (p == 0) ? cnt : n;
      const int nt = (int) ((m + os_tile_keys() - 1) / os_tile_keys());
      unsigned long long *hc = hist[p & 1], *hn = (p+1 < npass) ? hist[(p+1) & 1] : NULL;
      hipLaunchKernelGGL(os_scan256_kernel,dim3(1),dim3(256),0,dev->stream,hc,gbase,hn,ticket);
      const uint16_t *nov = NULL;
      // consecutive tiles one queue hands out: 7 x chunk tiles can be ahead of the smallest tile not yet taken and wait for
      // it, so chunk stays below 1/32 of the workgroups the device certainly holds at once (one per CU)
      unsigned chunk = (unsigned) (dev->ncu / 32);
      { static int forced = -1;
        if (forced < 0) { const char *e = getenv("FGA_SORT_CHUNK"); forced = e != NULL ? atoi(e) : 0; }
        if (forced > 0) chunk = (unsigned) forced;
      }
      if (chunk < 1) chunk = 1;
      if (chunk > 64) chunk = 64;
      if (os_wide() == 8)
        { if (p == 0 && from_seeds)
            hipLaunchKernelGGL((osw_pass_kernel<true,8,16>),dim3(nt),dim3(512),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,valid,chunk);
          else
            hipLaunchKernelGGL((osw_pass_kernel<false,8,16>),dim3(nt),dim3(512),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,nov,chunk);
        }
      else if (os_wide() == 16)
        { if (p == 0 && from_seeds)
            hipLaunchKernelGGL((osw_pass_kernel<true,16,KPT16>),dim3(nt),dim3(1024),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,valid,chunk);
          else
            hipLaunchKernelGGL((osw_pass_kernel<false,16,KPT16>),dim3(nt),dim3(1024),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,nov,chunk);
        }
      else if (p == 0 && from_seeds)
        hipLaunchKernelGGL(os_pass_kernel<true>,dim3(nt),dim3(ST),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,valid);
      else
        hipLaunchKernelGGL(os_pass_kernel<false>,dim3(nt),dim3(ST),0,dev->stream,src,dst,m,shift,shift+8,L,p+1,gbase,status,hn,ticket,
                           (const uint16_t *) NULL);
      src = dst;
      dst = (dst == buf0) ? buf1 : buf0;
    }
  *sorted = (uint4 *) src;
}

// FGA_SORT_3N=1: the three-kernel passes (tile histogram, scan, scatter: 3 n traffic per pass) instead of the one-sweep ones
static bool sort_three_kernel_passes()
{ static int v = -1;
  if (v < 0)
    { const char *e = getenv("FGA_SORT_3N");
      v = (e != NULL && atoi(e) != 0) ? 1 : 0;
    }
  return v != 0;
}

static int bits_for(int64_t maxval)      // bits needed to hold values 0..maxval
{ int b = 1;
  while ((maxval >> b) != 0) b++;
  return b;
}

extern "C" int fga_seed_sort(fga_dev *dev, const fga_dseeds *S, const fga_sort_params *prm, fga_dkeys **out)
{ *out = NULL;
  if (dev == NULL || S == NULL || prm == NULL)
    { fga_set_error("fga_seed_sort: null argument");
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  const int64_t next = fga_seeds_extent(S);                                   // slots of the seed buffer to look at
  const int64_t n = S->valid != NULL ? S->count : next;                        // keys that come out
  key_layout L;
  L.amx = prm->amxpos; L.bmx = prm->bmxpos;
  L.wa = bits_for(prm->nctg_a > 0 ? prm->nctg_a-1 : 0);
  L.wb = bits_for(prm->nctg_b > 0 ? prm->nctg_b-1 : 0);
  L.wt = bits_for(prm->amxpos + prm->bmxpos);
  L.wd = bits_for((prm->amxpos + prm->bmxpos) >> 6);
  const int tbits = 12 + L.wt + L.wd + L.wb + L.wa + 1;
  if (tbits > 128)
    { fga_set_error("fga_seed_sort: key of %d bits does not fit 128",tbits);
      return 1;
    }
  fga_dkeys *K = (fga_dkeys *) calloc(1,sizeof(fga_dkeys));
  if (K == NULL)
    { fga_set_error("out of memory");
      return 1;
    }
  K->dev = dev; K->count = n;
  K->wa = L.wa; K->wb = L.wb; K->wd = L.wd; K->wt = L.wt;
  K->amxpos = prm->amxpos; K->bmxpos = prm->bmxpos;
  const int lowbit = prm->anti_order_only ? 12 : 0;       // diag&63 and lcp: the chain scan does not need them ordered
  const int npass = (tbits - lowbit + 7) / 8;
  // the first pass walks the seed buffer (holes included), the others the keys
  const int ntiles0 = (int) ((next + STILE - 1) / STILE), ntilesk = (int) ((n + STILE - 1) / STILE);
  const int ntiles = ntiles0 > ntilesk ? ntiles0 : ntilesk;
  dev->last_ms[FGA_STAGE_SORT] = 0.f;
  if (n == 0)
    { *out = K;
      return 0;
    }
  uint4 *buf[2] = { NULL, NULL };
  uint32_t *hist = NULL, *sums = NULL;
  const int64_t hm = (int64_t) 256*ntiles;
  const int nch = (int) ((hm + SCAN_CH - 1) / SCAN_CH);
  hipError_t e;
  const bool three = sort_three_kernel_passes();
  const int64_t ostiles = ((next > n ? next : n) + OS_TILE - 1) / OS_TILE;
  const size_t wbytes = three ? sizeof(uint32_t)*(256*(size_t) ntiles + nch + 1) : os_work_bytes(ostiles);
  K->alloc_bytes = sizeof(uint4)*(size_t) n;
  e = hipSuccess;
  buf[0] = (uint4 *) fga_dev_acquire(dev,SLOT_SORT0,K->alloc_bytes);
  buf[1] = (uint4 *) fga_dev_acquire(dev,SLOT_SORT1,K->alloc_bytes);
  hist   = (uint32_t *) fga_dev_acquire(dev,SLOT_HIST,wbytes);
  if (buf[0] == NULL || buf[1] == NULL || hist == NULL)
    { fga_set_error("fga_seed_sort: device allocation failed");
      fga_dev_release(dev,SLOT_SORT0,buf[0]); fga_dev_release(dev,SLOT_SORT1,buf[1]);
      fga_dev_release(dev,SLOT_HIST,hist); free(K);
      return 1;
    }
  sums = hist + 256*(size_t) ntiles;
  hipEventRecord(dev->ev0,dev->stream);
  const void *src = S->seeds;
  if (!three)
    { uint4 *sorted = NULL;
      os_sort(dev,S->seeds,true,next,S->valid,L,buf[0],buf[1],n,lowbit,npass,hist,ostiles,&sorted);
      src = sorted;
    }
  else
  { int cur = 0;
  for (int p = 0; p < npass; p++)
    { const int shift = lowbit + 8*p;
      uint4 *dst = buf[cur];
      if (p == 0)
        { const int64_t hm0 = (int64_t) 256*ntiles0;
          const int nch0 = (int) ((hm0 + SCAN_CH - 1) / SCAN_CH);
          hipLaunchKernelGGL(sort_hist_kernel<true>,dim3(ntiles0),dim3(ST),0,dev->stream,src,next,shift,L,hist,ntiles0,S->valid);
          hipLaunchKernelGGL(sort_scan_local_kernel,dim3(nch0),dim3(SCAN_T),0,dev->stream,hist,hm0,sums);
          hipLaunchKernelGGL(sort_scan_sums_kernel,dim3(1),dim3(SCAN_T),0,dev->stream,sums,nch0);
          hipLaunchKernelGGL(sort_scatter_kernel<true>,dim3(ntiles0),dim3(ST),0,dev->stream,src,dst,next,shift,L,hist,sums,ntiles0,
                             S->valid);
        }
      else
        { const int64_t hmk = (int64_t) 256*ntilesk;
          const int nchk = (int) ((hmk + SCAN_CH - 1) / SCAN_CH);
          hipLaunchKernelGGL(sort_hist_kernel<false>,dim3(ntilesk),dim3(ST),0,dev->stream,src,n,shift,L,hist,ntilesk,
                             (const uint16_t *) NULL);
          hipLaunchKernelGGL(sort_scan_local_kernel,dim3(nchk),dim3(SCAN_T),0,dev->stream,hist,hmk,sums);
          hipLaunchKernelGGL(sort_scan_sums_kernel,dim3(1),dim3(SCAN_T),0,dev->stream,sums,nchk);
          hipLaunchKernelGGL(sort_scatter_kernel<false>,dim3(ntilesk),dim3(ST),0,dev->stream,src,dst,n,shift,L,hist,sums,ntilesk,
                             (const uint16_t *) NULL);
        }
      src = dst;
      cur ^= 1;
    }
  }
  hipEventRecord(dev->ev1,dev->stream);
  e = hipStreamSynchronize(dev->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess)
    { fga_set_error("fga_seed_sort: kernel failed: %s",hipGetErrorString(e));
      fga_dev_release(dev,SLOT_SORT0,buf[0]); fga_dev_release(dev,SLOT_SORT1,buf[1]);
      fga_dev_release(dev,SLOT_HIST,hist); free(K);
      return 1;
    }
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_SORT],dev->ev0,dev->ev1);
  K->keys = (uint4 *) src;
  K->slot = (src == buf[0]) ? SLOT_SORT0 : SLOT_SORT1;
  if (src == buf[0]) fga_dev_release(dev,SLOT_SORT1,buf[1]); else fga_dev_release(dev,SLOT_SORT0,buf[0]);
  fga_dev_release(dev,SLOT_HIST,hist);
  *out = K;
  return 0;
}

// LSD radix sort of n 128-bit keys on bits [lowbit, lowbit+nbits) with the kernels above; buf0 holds the input, buf1 is
// scratch of the same size; *sorted is whichever of the two holds the result.  Enqueued on dev->stream, not synchronised.
int fga_radix_sort_u128(fga_dev *dev, uint4 *buf0, uint4 *buf1, int64_t n, int lowbit, int nbits, uint4 **sorted)
{ *sorted = buf0;
  if (n <= 1 || nbits <= 0)
    return 0;
  const int npass = (nbits + 7) / 8;
  const int ntiles = (int) ((n + STILE - 1) / STILE);
  const int64_t hm = (int64_t) 256*ntiles;
  const int nch = (int) ((hm + SCAN_CH - 1) / SCAN_CH);
  const bool three = sort_three_kernel_passes();
  uint32_t *hist = (uint32_t *) fga_dev_acquire(dev,SLOT_HIST,three ? sizeof(uint32_t)*(256*(size_t) ntiles + nch + 1)
                                                                    : os_work_bytes((n + OS_TILE - 1) / OS_TILE));
  if (hist == NULL)
    { fga_set_error("radix sort: device allocation failed");
      return 1;
    }
  uint32_t *sums = hist + 256*(size_t) ntiles;
  key_layout L;
  memset(&L,0,sizeof(L));
  uint4 *src = buf0, *dst = buf1;
  if (!three)
    os_sort(dev,buf0,false,n,(const uint16_t *) NULL,L,buf1,buf0,n,lowbit,npass,hist,(n + OS_TILE - 1) / OS_TILE,&src);
  else
  for (int p = 0; p < npass; p++)
    { const int shift = lowbit + 8*p;
      hipLaunchKernelGGL(sort_hist_kernel<false>,dim3(ntiles),dim3(ST),0,dev->stream,(const void *) src,n,shift,L,hist,ntiles,
                         (const uint16_t *) NULL);
      hipLaunchKernelGGL(sort_scan_local_kernel,dim3(nch),dim3(SCAN_T),0,dev->stream,hist,hm,sums);
      hipLaunchKernelGGL(sort_scan_sums_kernel,dim3(1),dim3(SCAN_T),0,dev->stream,sums,nch);
      hipLaunchKernelGGL(sort_scatter_kernel<false>,dim3(ntiles),dim3(ST),0,dev->stream,(const void *) src,dst,n,shift,L,hist,sums,ntiles,
                         (const uint16_t *) NULL);
      uint4 *t = src; src = dst; dst = t;
    }
  // the histogram buffer is still in use by the enqueued kernels: the slot is grow-only memory of the device context and
  // the next user is ordered behind them on the same stream, so releasing the bookkeeping here is safe
  fga_dev_release(dev,SLOT_HIST,hist);
  *sorted = src;
  return 0;
}

// rmsd_sort on records that already sit in HBM (include/fastga_amd.h); what tools/ubench/sort_bench.hip times
extern "C" int fga_dev_radix_sort_u128(fga_dev *dev, void *buf0, void *buf1, int64_t n, int lowbit, int nbits, void **sorted)
{ if (dev == NULL || buf0 == NULL || buf1 == NULL || sorted == NULL || n < 0 || lowbit < 0 || nbits < 0 || lowbit + nbits > 128)
    { fga_set_error("fga_dev_radix_sort_u128: bad argument");
      return 1;
    }
  FGA_HIP(fga_dev_enter(dev));
  uint4 *res = NULL;
  hipEventRecord(dev->ev0,dev->stream);
  if (fga_radix_sort_u128(dev,(uint4 *) buf0,(uint4 *) buf1,n,lowbit,nbits,&res))
    return 1;
  hipEventRecord(dev->ev1,dev->stream);
  FGA_HIP(hipStreamSynchronize(dev->stream));
  FGA_HIP(hipGetLastError());
  hipEventElapsedTime(&dev->last_ms[FGA_STAGE_SORT],dev->ev0,dev->ev1);
  *sorted = res;
  return 0;
}

extern "C" int64_t fga_keys_count(const fga_dkeys *K) { return K->count; }

extern "C" int fga_keys_download(const fga_dkeys *K, void *host, int64_t max)
{ int64_t n = K->count < max ? K->count : max;
  FGA_HIP(fga_dev_enter(K->dev));
  if (n > 0)
    FGA_HIP(hipMemcpy(host,K->keys,sizeof(uint4)*(size_t) n,hipMemcpyDeviceToHost));
  return 0;
}

extern "C" void fga_keys_layout(const fga_dkeys *K, int *wa, int *wb, int *wd, int *wt)
{ *wa = K->wa; *wb = K->wb; *wd = K->wd; *wt = K->wt; }

extern "C" void fga_keys_free(fga_dkeys *K)
{ if (K == NULL) return;
  fga_dev_enter(K->dev);
  fga_dev_release(K->dev,K->slot,K->keys);
  free(K);
}

// zero-copy variant for the pipeline: keys land in the device context's pinned staging buffer
extern "C" const void *fga_keys_download_pinned(const fga_dkeys *K)
{ if (fga_dev_enter(K->dev) != hipSuccess) return NULL;
  void *h = fga_dev_pinned(K->dev,sizeof(uint4)*(size_t) (K->count+1));
  if (h == NULL)
    { fga_set_error("fga_keys_download_pinned: cannot allocate %lld bytes of pinned memory",
                    (long long) (sizeof(uint4)*(K->count+1)));
      return NULL;
    }
  if (K->count > 0 &&
      hipMemcpy(h,K->keys,sizeof(uint4)*(size_t) K->count,hipMemcpyDeviceToHost) != hipSuccess)
    { fga_set_error("fga_keys_download_pinned: copy failed");
      return NULL;
    }
  return h;
}

// ---------------------------------------------------------------------------------------------------
// exact-signature shim of rmsd_sort (declared FastGA.c:149-150, defined RSDsort.c:292; called once per (strand, part) at
// FastGA.c:4320): `array` holds nparts consecutive panels of part[p] BYTES each, made of rsize-byte records; every
// panel is sorted ascending on the ksize bytes at the END of the record, last byte most significant (the record read
// backwards); range[] receives the panel ranges the reference's sort threads would have been given -- the search threads
// reuse them (FastGA.c:4336-4345) -- and the number of ranges in use is returned.  The records go to the device as
// 128-bit keys [panel | record], one LSD radix sort orders all panels at once (fga_radix_sort_u128), and the low rsize
// bytes come back.  A parity device for A/B runs inside the unmodified pipeline; like the reference's, not re-entrant.
// Returns -1 on failure (message: fga_last_error).
// ---------------------------------------------------------------------------------------------------
typedef struct { int beg, end; int64_t off; } shim_range;       // = Range, FastGA.c:143-147 / RSDsort.c:254-258

static fga_dev *shim_sort_dev = NULL;

extern "C" int fga_shim_rmsd_sort(uint8_t *array, int64_t nelem, int rsize, int ksize, int nparts, int64_t *part,
                                  int nthreads, void *range_)
{ shim_range *range = (shim_range *) range_;
  if (array == NULL || part == NULL || range == NULL || rsize < 1 || ksize < 1 || ksize > rsize || nparts < 1 || nthreads < 1)
    { fga_set_error("fga_shim_rmsd_sort: bad argument");
      return -1;
    }
  int pbits = 1;
  while (((int64_t) 1 << pbits) < nparts) pbits += 1;
  if (8*rsize + pbits > 128)
    { fga_set_error("fga_shim_rmsd_sort: %d-byte records of %d panels do not fit a 128-bit key",rsize,nparts);
      return -1;
    }
  const int64_t asize = nelem*rsize;
  // the reference's thread ranges: consecutive panels of about asize/nthreads bytes each (RSDsort.c:318-343; fga_order.c)
  int n = 0;
  { std::vector<int> rb((size_t) nthreads), re((size_t) nthreads);
    std::vector<int64_t> ro((size_t) nthreads);
    n = fga_rmsd_ranges(part,nparts,asize,nthreads,rb.data(),re.data(),ro.data());
    for (int x = 0; x < nthreads; x++)
      { range[x].beg = rb[(size_t) x]; range[x].end = re[(size_t) x]; range[x].off = ro[(size_t) x]; }
  }
  if (nelem == 0)
    return n;
  if (shim_sort_dev == NULL)
    { const char *e = getenv("FGA_DEVICE");
      if (fga_dev_open(e != NULL ? atoi(e) : 0,&shim_sort_dev))
        return -1;
    }
  fga_dev *dev = shim_sort_dev;
  if (fga_dev_enter(dev) != hipSuccess)
    { fga_set_error("fga_shim_rmsd_sort: cannot select the device");
      return -1;
    }
  std::vector<uint4> keys((size_t) nelem);
  { int64_t i = 0;
    for (int p = 0; p < nparts; p++)
      for (int64_t b = 0; b < part[p] && i < nelem; b += rsize, i++)
        { unsigned __int128 k = 0;
          const uint8_t *r = array + i*rsize;
          for (int q = rsize-1; q >= 0; q--)
            k = (k << 8) | r[q];
          k |= (unsigned __int128) (unsigned) p << (8*rsize);
          keys[(size_t) i] = make_uint4((uint32_t) k,(uint32_t) (k >> 32),(uint32_t) (k >> 64),(uint32_t) (k >> 96));
        }
    if (i != nelem)
      { fga_set_error("fga_shim_rmsd_sort: the panel sizes do not add up to nelem records");
        return -1;
      }
  }
  uint4 *b0 = (uint4 *) fga_dev_acquire(dev,SLOT_SORT0,sizeof(uint4)*(size_t) nelem);
  uint4 *b1 = (uint4 *) fga_dev_acquire(dev,SLOT_SORT1,sizeof(uint4)*(size_t) nelem);
  uint4 *sorted = NULL;
  int rc = -1;
  if (b0 == NULL || b1 == NULL)
    fga_set_error("fga_shim_rmsd_sort: device allocation failed");
  else if (hipMemcpy(b0,keys.data(),sizeof(uint4)*(size_t) nelem,hipMemcpyHostToDevice) != hipSuccess)
    fga_set_error("fga_shim_rmsd_sort: upload failed");
  else if (fga_radix_sort_u128(dev,b0,b1,nelem,8*(rsize-ksize),8*ksize + pbits,&sorted) ||
           hipStreamSynchronize(dev->stream) != hipSuccess ||
           hipMemcpy(keys.data(),sorted,sizeof(uint4)*(size_t) nelem,hipMemcpyDeviceToHost) != hipSuccess)
    fga_set_error("fga_shim_rmsd_sort: device sort failed: %s",hipGetErrorString(hipGetLastError()));
  else
    { for (int64_t i = 0; i < nelem; i++)
        { const uint4 v = keys[(size_t) i];
          unsigned __int128 k = ((unsigned __int128) v.w << 96) | ((unsigned __int128) v.z << 64) |
                                ((unsigned __int128) v.y << 32) | v.x;
          uint8_t *r = array + i*rsize;
          for (int q = 0; q < rsize; q++, k >>= 8)
            r[q] = (uint8_t) k;
        }
      rc = n;
    }
  fga_dev_release(dev,SLOT_SORT0,b0); fga_dev_release(dev,SLOT_SORT1,b1);
  return rc;
}
