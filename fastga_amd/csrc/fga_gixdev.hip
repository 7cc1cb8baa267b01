// fga_gixdev.hip -- the genome index (GIX) built on the device, straight into HBM.
//
// Replaces GIXmake for the seed merge's input (reference GIXmake.c: sample / distribution / sort / merge threads and the
// .ktab / .gix writers): every 40-mer that starts with a closed (12,8)-syncmer under the TMap 4-mer order, both strands,
// sorted by (k-mer, payload), in the on-disk entry layout [7 suffix bytes | mask | lcp | post LE | contig|sign LE] with the
// int64[2^24] inclusive prefix index -- byte for byte what fga_gix_build (fga_gix.c, pinned against the reference's
// GIXmake) writes and fga_dgix_upload would load, so the merge kernels take it unchanged.  Nothing touches the disk:
// FASTA -> GDB (host parse) -> 2-bit image in HBM -> this -> seed merge.
//
//   gix_scan_kernel      one workgroup per 2048 positions of a contig: unpacks the bases into LDS, canonical 8-mer codes,
//                        sliding minimum over five codes -> syncmer starts; every (syncmer, strand) whose 40-mer fits the
//                        contig is one 128-bit key  [80-bit k-mer | payload].  Run twice: the first pass COUNTS -- per 12-mer
//                        prefix, and GIXmake's 5-base sample histogram (which decides the table parts, hence where the lcp
//                        byte restarts at 0) --, the second PLACES every key inside its prefix's panel of the final table
//                        (slot = end of the panel - what an atomic counter of the panel hands out): the reference's own
//                        MSD step (GIXmake.c distribution threads), the panels being known from the counts
//   gix_index_*_kernel   inclusive scan of the 2^24 prefix counts -> the index (between the two scans)
//   gix_tiles_kernel / gix_tile_sort_kernel   the panels are put in order in LDS: a workgroup takes the panels that start in
//                        one 1024-key stretch of the table (<= 2048 keys), spreads them over 256 buckets by the leading 8 bits
//                        in which the stretch's keys differ (an LDS counting pass) and ranks every key inside its bucket by
//                        comparison; panels beyond 2048 keys (low-complexity sequence) go to a list and through the seed sort's
//                        LSD radix passes.  13 global radix passes over all keys (0.45 s per 3 Gbp genome) are gone.
//   gix_entries_kernel   one thread per sorted key: lcp with its predecessor (0 at a part start), on-disk entry bytes
// What bounds the two scans (1 Gbp genome, 0.79 G keys: 33 + 43 ms): their atomics -- one per key on the 64 MB of panel counters,
// 25 G per second whatever else the kernel does (without the placing pass's returning atomic it takes 32 ms, without its
// 16-byte stores 33 ms; cutting every k-mer out of packed words instead of 40 byte reads changed 4 %).
#include "fga_device.hpp"

#define GCH   2048            // positions per chunk
#define GNT   256
#define GPER  (GCH/GNT)

__constant__ uint8_t gix_tmap[256];

struct gix_item { int ctg; int j0; };

struct gix_scan_args
  { const uint8_t *img;             // 2-bit image, contigs byte aligned, base i in bits 2*(i&3) of byte i>>2
    const int64_t *boff;            // byte offset of each contig
    const int64_t *clen;
    const int     *invp;            // original contig -> length-sorted index
    const gix_item *items; int nitems;
    int postbytes, contbytes;
    uint4 *keys;                       // NULL: the counting pass (per-prefix counts, the sample histogram, the number of keys);
    const int64_t *index;              // else the placing pass: key -> keys[index[prefix] - (what is left of count[prefix])]
    uint32_t pbeg, pend;               // only k-mers whose 12-mer prefix lies in [pbeg,pend) are kept (a rank's slice)
    const int64_t *goff;               // payloads beyond 6 bytes (do not fit under the k-mer in a 128-bit key): [ncontig+1] start of
                                       //   each ORIGINAL contig in a concatenation with gaps; the key then carries (global position
                                       //   << 1 | strand) in its 48 payload bits and gix_entries_kernel turns it back; else NULL
    unsigned long long *nkeys;
    uint32_t *count;                // [2^24]
    unsigned long long *sbuck;      // [1024]
  };

// The chunk's bases stay PACKED in LDS (2 bits each, base t of the chunk in bits 2*(t & 15) of dword t >> 4: the order of the
// .bps image) and every quantity is cut out of 64-bit windows of it with v_alignbit -- round 3 unpacked them into a byte per
// base and built every k-mer with 40 LDS byte reads, which was three quarters of the kernel's LDS instructions.
//   the forward 40-mer of position x, most significant base first   = the window of bases x .. x+39 with its 2-bit groups
//                                                                       reversed (v_bfrev + a swap of neighbouring bits)
//   the complement-strand 40-mer that ends at x+11                    = the bitwise NOT of the window of bases x-28 .. x+11
//                                                                       (its last base is the k-mer's first, complemented)
//   the 12-mer prefixes of both                                       = the 24 bits of bases x .. x+11, reversed / complemented
//   the canonical 8-mer code (TMap order)                             = two table lookups per strand on the window's bytes
__device__ __forceinline__ uint32_t rev2_32(uint32_t v)       // the sixteen 2-bit groups of v in reverse order
{ const uint32_t r = __builtin_bitreverse32(v);
  return ((r & 0xaaaaaaaau) >> 1) | ((r & 0x55555555u) << 1);
}

__device__ __forceinline__ uint64_t pw_window(const uint32_t *pw, int t)       // bases t .. t+31 of the chunk, base t in bits 0-1
{ const int w = t >> 4;
  const uint32_t sh = (uint32_t) (t & 15) * 2;
  const uint32_t d0 = pw[w], d1 = pw[w+1], d2 = pw[w+2];
  return ((uint64_t) __builtin_amdgcn_alignbit(d2,d1,sh) << 32) | __builtin_amdgcn_alignbit(d1,d0,sh);
}

#define GPRE  32              // bases in front of the chunk that are kept (a complement k-mer reaches 28 back)
#define GPW   ((GCH + 96)/16 + 3)

__global__ __launch_bounds__(GNT)
void gix_scan_kernel(gix_scan_args A)
{ __shared__ uint32_t pw[GPW];                // packed bases of [j0-GPRE, j0+GCH+64), zero outside the contig's bytes
  __shared__ uint16_t v8[GCH + 8];            // canonical 8-mer code at j0 + x
  __shared__ uint8_t  tf[256], tc[256];       // TMap code of a window byte (4 bases, first base in bits 0-1) read forward / as its reverse complement
  __shared__ uint32_t sh[1024];
  __shared__ int      wtot[GNT/64];

  const int tid = threadIdx.x;
  for (int x = tid; x < 1024; x += GNT)
    sh[x] = 0;
  { const uint32_t y = (uint32_t) tid;                                   // GNT == 256: one table entry per thread
    tf[y] = gix_tmap[rev2_32(y) >> 24];                                  // the 4-mer with its first base most significant
    tc[y] = gix_tmap[~y & 0xffu];                                        // complemented, last base first: the same bits, inverted
  }
  // A workgroup takes chunks in a stride and adds its sample histogram to the global one ONCE, at its end: a chunk holds
  // syncmers of ~600 of the 1024 sample buckets, and with one workgroup per chunk a 3 Gbp genome made 9 * 10^8 atomics on
  // the 128 cache lines of that histogram -- atomics on one line are served one after the other (~88 per microsecond).
  for (int item = blockIdx.x; item < A.nitems; item += gridDim.x)
  {
  __syncthreads();                            // the arrays of the chunk before are done with (and sh[] is cleared)
  const gix_item it = A.items[item];
  const int c = it.ctg, j0 = it.j0;
  const int64_t len = A.clen[c];
  const uint8_t *img = A.img + A.boff[c];
  { const int64_t nbytes = (len + 3) >> 2, b0 = ((int64_t) j0 - GPRE) >> 2;        // j0 is a multiple of GCH: whole bytes
    uint8_t *pb = (uint8_t *) pw;
    for (int y = tid; y < 4*GPW; y += GNT)
      { const int64_t cb = b0 + y;
        pb[y] = (cb >= 0 && cb < nbytes) ? img[cb] : (uint8_t) 0;
      }
  }
  __syncthreads();
  for (int x = tid; x < GCH + 8; x += GNT)
    { uint16_t v = 0xffff;
      if ((int64_t) j0 + x + 8 <= len)
        { const uint32_t y = (uint32_t) pw_window(pw,x + GPRE) & 0xffffu;          // bases x .. x+7
          const uint32_t ya = y & 0xffu, yb = y >> 8;
          const uint32_t mn = ((uint32_t) tf[ya] << 8) | tf[yb];
          const uint32_t mc = ((uint32_t) tc[yb] << 8) | tc[ya];
          v = (uint16_t) (mn < mc ? mn : mc);
        }
      v8[x] = v;
    }
  __syncthreads();

  // which of my positions start a syncmer, and which strands fit
  uint32_t fmask = 0, cmask = 0;
  int cnt = 0;
  #pragma unroll
  for (int r = 0; r < GPER; r++)
    { const int x = r*GNT + tid;
      const int64_t j = (int64_t) j0 + x;
      if (j + 12 > len)
        continue;
      uint16_t m = v8[x];
      #pragma unroll
      for (int q = 1; q <= 4; q++)
        m = v8[x+q] < m ? v8[x+q] : m;
      if (v8[x] != m && v8[x+4] != m)
        continue;
      // GIXmake's sample: every syncmer, both strands, whether or not the 40-mer fits
      // 12-mer prefix of the forward / complement k-mer of this syncmer: bases x .. x+11, first / last base most significant
      const uint32_t w24 = (uint32_t) pw_window(pw,x + GPRE) & 0xffffffu;
      const uint32_t f12 = rev2_32(w24) >> 8, c12 = ~w24 & 0xffffffu;
      if (A.keys == NULL)
        { atomicAdd(&sh[f12 >> 14],1u);
          atomicAdd(&sh[c12 >> 14],1u);
        }
      if (j <= len - FGA_KMER && f12 >= A.pbeg && f12 < A.pend)
        { fmask |= 1u << r; cnt += 1;
          if (A.keys == NULL) atomicAdd(A.count + f12,1u);
        }
      if (j >= FGA_KMER - 12 && c12 >= A.pbeg && c12 < A.pend)
        { cmask |= 1u << r; cnt += 1;
          if (A.keys == NULL) atomicAdd(A.count + c12,1u);
        }
    }

  if (A.keys == NULL)                               // counting pass: the number of keys, one global atomic per workgroup
    { int inc = cnt;
      for (int d = 1; d < 64; d <<= 1)
        inc += __shfl_xor(inc,d,64);
      if ((tid & 63) == 0)
        wtot[tid >> 6] = inc;
      __syncthreads();
      if (tid == 0)
        { int total = 0;
          for (int w = 0; w < GNT/64; w++) total += wtot[w];
          if (total > 0) atomicAdd(A.nkeys,(unsigned long long) total);
        }
      continue;
    }

  const uint64_t ctg = (uint64_t) A.invp[c];
  const uint64_t sign = 0x80ull << (8*(A.contbytes-1));
  #pragma unroll
  for (int r = 0; r < GPER; r++)
    { const int x = r*GNT + tid;
      const int64_t j = (int64_t) j0 + x;
      #pragma unroll
      for (int strand = 0; strand < 2; strand++)
        { if (!(((strand ? cmask : fmask) >> r) & 1))
            continue;
          uint64_t hi;
          uint32_t lo16;
          if (strand == 0)
            { const uint64_t v0 = pw_window(pw,x + GPRE);                               // bases x .. x+31
              const uint32_t v1 = (uint32_t) pw_window(pw,x + GPRE + 32) & 0xffffu;     //       x+32 .. x+39
              hi = ((uint64_t) rev2_32((uint32_t) v0) << 32) | rev2_32((uint32_t) (v0 >> 32));
              lo16 = rev2_32(v1) >> 16;
            }
          else
            { const uint64_t w0 = pw_window(pw,x + GPRE - 28);                          // bases x-28 .. x+3
              const uint64_t w1 = pw_window(pw,x + GPRE - 20);                          //       x-20 .. x+11: the k-mer's first 32 bases, last first
              hi = ~w1;
              lo16 = ~(uint32_t) w0 & 0xffffu;                                          // bases x-28 .. x-21: its last 8
            }
          uint64_t lo;
          if (A.goff != NULL)
            lo = ((uint64_t) lo16 << 48) | ((((uint64_t) A.goff[c] + (uint64_t) (strand ? j+12 : j)) << 1) | (uint64_t) strand);
          else
            { const uint64_t pay = strand ? ((uint64_t) (j+12) | ((ctg | sign) << (8*A.postbytes)))
                                          : ((uint64_t) j | (ctg << (8*A.postbytes)));
              lo = ((uint64_t) lo16 << 48) | (pay << (48 - 8*(A.postbytes+A.contbytes)));    // payload left-aligned under the k-mer: the significant key bits are contiguous
            }
          // into its panel: the panel [index[p] - count[p], index[p]) fills from the end (the order inside it is made below)
          const uint32_t p = (uint32_t) (hi >> 40);
          const uint32_t left = atomicSub(A.count + p,1u);
          A.keys[A.index[p] - (int64_t) left] = make_uint4((uint32_t) lo,(uint32_t) (lo >> 32),(uint32_t) hi,(uint32_t) (hi >> 32));
        }
    }
  }
  __syncthreads();
  if (A.keys == NULL)
    for (int x = tid; x < 1024; x += GNT)
      if (sh[x] != 0)
        atomicAdd(A.sbuck + x,(unsigned long long) sh[x]);
}

// ---- prefix counts -> inclusive int64 index (three small kernels over 2^24 counters) ----
#define ICH 4096
__global__ __launch_bounds__(256)
void gix_index_sums_kernel(const uint32_t *count, unsigned long long *sums)
{ __shared__ unsigned long long part[256];
  unsigned long long v = 0;
  const uint32_t *p = count + (size_t) blockIdx.x*ICH;
  for (int x = threadIdx.x; x < ICH; x += 256)
    v += p[x];
  part[threadIdx.x] = v;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1)
    { if ((int) threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x+d];
      __syncthreads();
    }
  if (threadIdx.x == 0)
    sums[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(1024)
void gix_index_scan_kernel(unsigned long long *sums, int n, unsigned long long *maxpre_unused)     // n = 4096: exclusive scan in place
{ __shared__ unsigned long long t[4096];
  for (int x = threadIdx.x; x < n; x += 1024) t[x] = sums[x];
  __syncthreads();
  if (threadIdx.x == 0)
    { unsigned long long run = 0;
      for (int x = 0; x < n; x++) { const unsigned long long v = t[x]; t[x] = run; run += v; }
    }
  __syncthreads();
  for (int x = threadIdx.x; x < n; x += 1024) sums[x] = t[x];
  (void) maxpre_unused;
}

__global__ __launch_bounds__(256)
void gix_index_write_kernel(const uint32_t *count, const unsigned long long *sums, int64_t *index, unsigned int *maxpre)
{ __shared__ unsigned long long part[256];
  const uint32_t *p = count + (size_t) blockIdx.x*ICH;
  int64_t *o = index + (size_t) blockIdx.x*ICH;
  // thread t owns 16 consecutive counters
  unsigned long long v = 0;
  uint32_t c[16], mx = 0;
  #pragma unroll
  for (int q = 0; q < 16; q++)
    { c[q] = p[threadIdx.x*16 + q]; v += c[q]; mx = c[q] > mx ? c[q] : mx; }
  part[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    { unsigned long long run = sums[blockIdx.x];
      for (int x = 0; x < 256; x++) { const unsigned long long w = part[x]; part[x] = run; run += w; }
    }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  #pragma unroll
  for (int q = 0; q < 16; q++)
    { run += c[q]; o[threadIdx.x*16 + q] = (int64_t) run; }
  if (mx > 0)
    atomicMax(maxpre,mx);
}

// ---- the panels put in order: LDS sort of the panels that start in one stretch of the table ----
#define GT_T    1024          // a tile: the panels that START in [b*GT_T, (b+1)*GT_T) of the table
#define GT_CAP  2048          // keys one workgroup orders in LDS at a time (every panel of a tile but the last ends inside the stretch)
#define GT_NT   256

struct gix_tile { int64_t lo, last; uint32_t pfirst, plast; };    // first key of the tile; first key and prefix of its last panel; prefix of its first
struct gix_over { int64_t start, count; };                        // a panel beyond GT_CAP keys

__device__ __forceinline__ int64_t gix_excl(const int64_t *index, int64_t p) { return p <= 0 ? 0 : index[p-1]; }

// tile b = the panels that start in [b*GT_T, (b+1)*GT_T): tiles[b].lo = start of the first of them (tiles[b+1].lo is where the
// tile ends; tiles[ntiles].lo = n), the last non-empty one and the prefixes the tile spans
__global__ __launch_bounds__(256)
void gix_tiles_kernel(const int64_t *index, int64_t n, int64_t ntiles, gix_tile *tiles)
{ const int64_t b = (int64_t) blockIdx.x*256 + threadIdx.x;
  if (b > ntiles)
    return;
  int64_t lo[2];
  for (int k = 0; k < 2; k++)                        // smallest p in [0, 2^24] with excl(p) >= (b+k)*GT_T
    { const int64_t v = (b+k)*GT_T;
      int64_t a = 0, e = FGA_NPREFIX;
      while (a < e)
        { const int64_t m = (a+e) >> 1;
          if (gix_excl(index,m) >= v) e = m; else a = m+1;
        }
      lo[k] = gix_excl(index,a);
    }
  gix_tile t;
  t.lo = lo[0]; t.last = lo[0]; t.pfirst = t.plast = 0;
  if (lo[1] > lo[0])
    { int64_t a = 0, e = FGA_NPREFIX-1;               // the last panel: smallest p with index[p] >= lo[1]
      while (a < e)
        { const int64_t m = (a+e) >> 1;
          if (index[m] >= lo[1]) e = m; else a = m+1;
        }
      t.plast = (uint32_t) a; t.last = gix_excl(index,a);
      a = 0; e = FGA_NPREFIX-1;                       // the first one: smallest p with index[p] > lo[0]
      while (a < e)
        { const int64_t m = (a+e) >> 1;
          if (index[m] > lo[0]) e = m; else a = m+1;
        }
      t.pfirst = (uint32_t) a;
    }
  tiles[b] = t;
}

__device__ __forceinline__ bool gix_key_less(const uint4 &a, const uint4 &b)
{ const uint64_t ah = ((uint64_t) a.w << 32) | a.z, bh = ((uint64_t) b.w << 32) | b.z;
  const uint64_t al = ((uint64_t) a.y << 32) | a.x, bl = ((uint64_t) b.y << 32) | b.x;
  return ah < bh || (ah == bh && al < bl);
}

// keys[0..m) (m <= GT_CAP) of the prefixes [pfirst, plast] into ascending order.  One counting pass in LDS on the leading 8
// bits in which keys of these prefixes can differ (the prefix bits that vary, then the first bases of the suffix), then every
// key's rank inside its bucket by comparison with the bucket's other keys (a handful; the lanes of a wavefront sit in the
// same or neighbouring buckets and read the same LDS words).  Keys are distinct (the payload is a position).
__device__ void gix_lds_sort(uint4 *keys, int m, uint32_t pfirst, uint32_t plast, uint4 *bufA, uint4 *bufB,
                             uint32_t *cnt, uint32_t *bs, uint32_t *fill, uint32_t *wsum)
{ const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t span = plast - pfirst;
  const int sh = 32 + (span == 0 ? 0 : 32 - __clz((int) span));       // (hi - (pfirst << 40)) >> sh < 256
  const uint64_t base = (uint64_t) pfirst << 40;
  cnt[tid] = 0;
  __syncthreads();
  for (int x = tid; x < m; x += GT_NT)
    { const uint4 k = keys[x];
      bufA[x] = k;
      const uint64_t hi = ((uint64_t) k.w << 32) | k.z;
      atomicAdd(&cnt[(uint32_t) ((hi - base) >> sh) & 255u],1u);
    }
  __syncthreads();
  { const uint32_t c = cnt[tid];
    uint32_t inc = c;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1)
      { const uint32_t y = __shfl_up(inc,d,64);
        if (lane >= d) inc += y;
      }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; w++) off += wsum[w];
    bs[tid] = off + inc - c; fill[tid] = off + inc - c;
    if (tid == GT_NT-1) bs[GT_NT] = off + inc;
  }
  __syncthreads();
  for (int x = tid; x < m; x += GT_NT)
    { const uint4 k = bufA[x];
      const uint64_t hi = ((uint64_t) k.w << 32) | k.z;
      const uint32_t at = atomicAdd(&fill[(uint32_t) ((hi - base) >> sh) & 255u],1u);
      bufB[at] = k;
    }
  __syncthreads();
  for (int x = tid; x < m; x += GT_NT)
    { const uint4 k = bufB[x];
      const uint64_t hi = ((uint64_t) k.w << 32) | k.z;
      const uint32_t d = (uint32_t) ((hi - base) >> sh) & 255u;
      const int s = (int) bs[d], e = (int) bs[d+1];
      int r = 0;
      for (int j = s; j < e; j++)
        { const uint4 q = bufB[j];
          r += (gix_key_less(q,k) || (j < x && !gix_key_less(k,q))) ? 1 : 0;
        }
      bufA[s + r] = k;
    }
  __syncthreads();
  for (int x = tid; x < m; x += GT_NT)
    keys[x] = bufA[x];
  __syncthreads();
}

__global__ __launch_bounds__(GT_NT)
void gix_tile_sort_kernel(uint4 *keys, const gix_tile *tiles, gix_over *over, unsigned int *nover)
{ __shared__ uint4 bufA[GT_CAP], bufB[GT_CAP];
  __shared__ uint32_t cnt[GT_NT], bs[GT_NT+1], fill[GT_NT], wsum[GT_NT/64];
  const gix_tile t = tiles[blockIdx.x];
  const int64_t hi = tiles[blockIdx.x+1].lo, m = hi - t.lo;
  if (m <= 1)
    return;
  if (m <= GT_CAP)
    { gix_lds_sort(keys + t.lo,(int) m,t.pfirst,t.plast,bufA,bufB,cnt,bs,fill,wsum);
      return;
    }
  // the last panel reaches beyond the stretch: the panels before it (they end inside it: < GT_T keys), then the last one
  if (t.last - t.lo > 1)
    gix_lds_sort(keys + t.lo,(int) (t.last - t.lo),t.pfirst,t.plast > t.pfirst ? t.plast-1 : t.pfirst,bufA,bufB,cnt,bs,fill,wsum);
  const int64_t s = hi - t.last;
  if (s <= GT_CAP)
    gix_lds_sort(keys + t.last,(int) s,t.plast,t.plast,bufA,bufB,cnt,bs,fill,wsum);
  else if (threadIdx.x == 0)
    { const unsigned int k = atomicAdd(nover,1u);
      over[k].start = t.last; over[k].count = s;
    }
}

// the panels beyond GT_CAP keys, gathered into one stretch (dir 0) / put back (dir 1): off[k] = first slot of panel k
__global__ __launch_bounds__(256)
void gix_over_copy_kernel(uint4 *keys, uint4 *scratch, const gix_over *over, const int64_t *off, int nover, int64_t total, int dir)
{ for (int64_t i = (int64_t) blockIdx.x*256 + threadIdx.x; i < total; i += (int64_t) gridDim.x*256)
    { int a = 0, e = nover-1;                          // largest k with off[k] <= i
      while (a < e)
        { const int m = (a+e+1) >> 1;
          if (off[m] <= i) a = m; else e = m-1;
        }
      const int64_t src = over[a].start + (i - off[a]);
      if (dir == 0) scratch[i] = keys[src];
      else          keys[src] = scratch[i];
    }
}

// ---- sorted keys -> on-disk entries ----
struct gix_entries_args
  { const uint4 *keys; int64_t n;
    int postbytes, contbytes, ebytes;
    uint8_t *table;                 // TO_VIEW == false: the on-disk entry bytes
    fga_view view;                  // TO_VIEW == true: the merge kernel's field arrays, written directly (fga_view.hip)
    const uint8_t *partid;          // [1024] number of part boundaries at or below this 5-base bucket
    // soft mask (optional): lower-case intervals per original contig, and the sorted -> original contig map
    const int64_t *moff, *mbeg, *mend;
    const int     *perm;
    const int64_t *goff; int ngoff;  // wide payloads: the keys carry global positions (gix_scan_args.goff); ngoff = contigs
    const int     *invp;             //   original contig -> length-sorted index
  };

template <bool TO_VIEW>
__device__ __forceinline__ void gix_entry_one(const gix_entries_args &A, const int64_t i);

// one entry per thread and turn (a dispatch holds at most 2^32 work-items: a table of more entries takes several turns)
template <bool TO_VIEW>
__global__ __launch_bounds__(256)
void gix_entries_kernel(gix_entries_args A)
{ for (int64_t i = (int64_t) blockIdx.x*blockDim.x + threadIdx.x; i < A.n; i += (int64_t) gridDim.x*blockDim.x)
    gix_entry_one<TO_VIEW>(A,i);
}

template <bool TO_VIEW>
__device__ __forceinline__ void gix_entry_one(const gix_entries_args &A, const int64_t i)
{
  const uint4 k = A.keys[i];
  const uint64_t hi = ((uint64_t) k.w << 32) | k.z, lo = ((uint64_t) k.y << 32) | k.x;
  const uint32_t lo16 = (uint32_t) (lo >> 48);
  int lcp = 0;
  if (i > 0)
    { const uint4 q = A.keys[i-1];
      const uint64_t phi = ((uint64_t) q.w << 32) | q.z;
      const uint32_t plo16 = (uint32_t) ((((uint64_t) q.y << 32) | q.x) >> 48);
      if (A.partid[hi >> 54] != A.partid[phi >> 54])
        lcp = 0;                                       // first entry of a table part
      else if (hi != phi)
        lcp = __clzll((long long) (hi ^ phi)) >> 1;
      else if (lo16 != plo16)
        lcp = 32 + ((__clz((int) (lo16 ^ plo16)) - 16) >> 1);
      else
        lcp = FGA_KMER;
    }
  const uint64_t suf = ((hi & 0xffffffffffull) << 16) | lo16;           // bases 12..39
  uint64_t pay;
  if (A.goff != NULL)
    { const uint64_t code = lo & 0xffffffffffffull;
      const int64_t g = (int64_t) (code >> 1);
      int a = 0, b = A.ngoff;                          // the contig whose stretch holds g: largest c with goff[c] <= g
      while (b - a > 1)
        { const int m = (a + b) >> 1;
          if (A.goff[m] <= g) a = m; else b = m;
        }
      const uint64_t cw = (uint64_t) A.invp[a] | ((code & 1) ? (0x80ull << (8*(A.contbytes-1))) : 0ull);
      pay = (uint64_t) (g - A.goff[a]) | (cw << (8*A.postbytes));
    }
  else
    pay = (lo & 0xffffffffffffull) >> (48 - 8*(A.postbytes+A.contbytes));
  uint8_t *o = TO_VIEW ? NULL : A.table + (size_t) i*A.ebytes;
  if (!TO_VIEW)
    { 
      #pragma unroll
      for (int q = 0; q < 7; q++)
        o[q] = (uint8_t) (suf >> (8*(6-q)));
    }
  // soft-mask byte of the k-mers whose syncmer starts at j: bases from j to the end of the lower-case interval that
  // holds j, capped at 40; 0 outside intervals (setup_thread_with_masks, GIXmake.c:1100-1108)
  uint32_t pbg = 0;
  if (A.moff != NULL)
    { const uint64_t pm = (A.postbytes >= 8) ? ~0ull : ((1ull << (8*A.postbytes)) - 1);
      const uint64_t cw = pay >> (8*A.postbytes);
      const uint64_t sb = 0x80ull << (8*(A.contbytes-1));
      const int c = A.perm[(int) (cw & (sb-1))];
      const int64_t j = (int64_t) (pay & pm) - ((cw & sb) ? 12 : 0);
      int64_t lo_ = A.moff[c], hi_ = A.moff[c+1];            // first interval with mend > j
      while (lo_ < hi_)
        { const int64_t m = (lo_ + hi_) >> 1;
          if (A.mend[m] > j) hi_ = m; else lo_ = m+1;
        }
      if (lo_ < A.moff[c+1] && j >= A.mbeg[lo_])
        { const int64_t dd = A.mend[lo_] - j;
          pbg = (uint32_t) (dd > FGA_KMER ? FGA_KMER : dd);
        }
    }
  if (TO_VIEW)
    { // what view_repack_kernel makes of the entry bytes: the key carries the low byte of the 12-mer prefix; the first entry
      // of a panel has an lcp below 12 by construction (its 12-mer differs from the one before)
      const fga_view &V = A.view;
      const uint32_t pm = A.postbytes >= 4 ? 0xffffffffu : ((1u << (8*A.postbytes)) - 1);
      V.K[i] = (((hi >> 40) & 0xff) << 56) | suf;
      V.L[i] = (uint8_t) lcp;
      V.M[i] = (uint8_t) pbg;
      V.P[i] = (uint32_t) pay & pm;
      const uint32_t c = (uint32_t) (pay >> (8*A.postbytes)) & (A.contbytes >= 4 ? 0xffffffffu : ((1u << (8*A.contbytes)) - 1));
      if (V.cw == 1)      ((uint8_t  *) V.C)[i] = (uint8_t) c;
      else if (V.cw == 2) ((uint16_t *) V.C)[i] = (uint16_t) c;
      else                ((uint32_t *) V.C)[i] = c;
      return;
    }
  o[7] = (uint8_t) pbg;
  o[8] = (uint8_t) lcp;
  for (int q = 0; q < A.postbytes + A.contbytes; q++)
    o[9+q] = (uint8_t) (pay >> (8*q));
}

// the view's 32-bit prefix index from the 64-bit one
__global__ __launch_bounds__(256)
void gix_view_index_kernel(const int64_t *idx64, uint32_t *idx32)
{ const int64_t p = (int64_t) blockIdx.x*256 + threadIdx.x;
  if (p < FGA_NPREFIX)
    idx32[p] = (uint32_t) idx64[p];
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// the build proper.  [pbeg,pend): the 12-mer prefixes kept (a rank's slice of the table; the whole space for the whole
// table).  counts_host != NULL: count only -- the per-prefix entry counts of the whole table come back, nothing is built.
// keep_img != NULL: the genome's bases stay on the device after the build -- the image is laid out as fga_dgenome wants it
// (FGA_IMG_PAD zero bytes either side) and handed to the caller, who gives it to fga_dgenome_adopt: the bases cross PCIe
// once per session, not twice (0.75 GB per 3 Gbp genome from pageable memory: ~0.1 s of the 0.45 s a genome costs to open)
static int dgix_build_impl(fga_dev *dev, const fga_gdb *G, int nthreads, int flags, int64_t pbeg, int64_t pend,
                           uint32_t *counts_host, fga_dgix **dout, fga_gix **xout, uint8_t **keep_img)
{ if (dout != NULL) *dout = NULL;
  if (keep_img != NULL) *keep_img = NULL;
  if (xout != NULL) *xout = NULL;
  const bool count_only = counts_host != NULL;
  if (pbeg < 0) pbeg = 0;
  if (pend > FGA_NPREFIX || pend <= 0) pend = FGA_NPREFIX;
  const int want_host_copy = (flags & FGA_GIX_HOST_COPY) != 0;
  bool direct_view = false;
  const int use_mask = (flags & FGA_GIX_SOFT_MASK) != 0 && G->nmask > 0;
  int64_t *dmoff = NULL, *dmbeg = NULL, *dmend = NULL;
  int *dperm = NULL;
  FGA_HIP(fga_dev_enter(dev));
  int nctg = 0, postbytes = 0, contbytes = 0, nparts = 0;
  int *perm = NULL, *invp = NULL;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 32) nthreads = 32;
  if (fga_gix_layout(G,nthreads,&nctg,&perm,&invp,&postbytes,&contbytes,&nparts))
    return 1;
  const int ebytes = 9 + postbytes + contbytes;
  int status = 1;
  fga_dgix *D = NULL;
  fga_gix  *X = NULL;
  uint8_t *dimg = NULL, *dpartid = NULL, *dimg0 = NULL;      // dimg0: the padded allocation, dimg = dimg0 + FGA_IMG_PAD
  int64_t *dboff = NULL, *dclen = NULL;
  int *dinvp = NULL;
  gix_item *ditems = NULL;
  uint32_t *dcount = NULL;
  unsigned long long *dctr = NULL;       // [0] keys, [1..1024] sbuck, then 4096 chunk sums, then maxpre
  uint4 *buf0 = NULL, *sorted = NULL, *oscr = NULL;
  gix_tile *dtiles = NULL;
  gix_over *dover = NULL;
  int64_t *dooff = NULL, *dgoff = NULL;
  unsigned scan_grid = 1;                  // workgroups of the scan kernel (each takes chunks in a stride)
  std::vector<gix_item> items;
  std::vector<int64_t> boff((size_t) G->ncontig), clen((size_t) G->ncontig);
  int64_t nkeys = 0, ntiles = 0;
  hipError_t e = hipSuccess;
  float ms = 0.f;
  double tn = 0.;

  // payloads of 7 and 8 bytes (more than 32,768 contigs with one beyond 16.7 Mbp: GIXmake.c:1888-1901 sizes them freely) do
  // not fit under the k-mer in the 128-bit key: the key then carries the k-mer's position in a concatenation of the contigs
  const bool wide_pay = postbytes + contbytes > 6;
  std::vector<int64_t> goff;
  if (postbytes + contbytes > 8 || postbytes > 4)
    { fga_set_error("fga_dgix_build: %d position and %d contig bytes: payloads beyond 8 bytes / contigs beyond 4 Gbp are not supported",
                    postbytes,contbytes);
      goto done;
    }
  if (wide_pay)
    { goff.resize((size_t) G->ncontig + 1);
      int64_t run = 0;
      for (int c = 0; c < G->ncontig; c++)
        { goff[(size_t) c] = run; run += G->contigs[c].clen + 64; }
      goff[(size_t) G->ncontig] = run;
      if (run >= ((int64_t) 1 << 46))
        { fga_set_error("fga_dgix_build: genome too large for the position code of wide payloads");
          goto done;
        }
    }
  for (int c = 0; c < G->ncontig; c++)
    { boff[(size_t) c] = G->contigs[c].boff; clen[(size_t) c] = G->contigs[c].clen;
      for (int64_t j = 0; j + 12 <= G->contigs[c].clen; j += GCH)
        { gix_item it; it.ctg = c; it.j0 = (int) j; items.push_back(it); }
    }
  if (items.empty())
    { fga_set_error("fga_dgix_build: no contig is long enough to hold a syncmer");
      goto done;
    }

  if ((e = fga_dmalloc(&dimg0,(size_t) G->bpslen + 2*FGA_IMG_PAD + 64)) != hipSuccess ||
      (e = fga_dmalloc(&dboff,sizeof(int64_t)*boff.size())) != hipSuccess ||
      (e = fga_dmalloc(&dclen,sizeof(int64_t)*clen.size())) != hipSuccess ||
      (e = fga_dmalloc(&dinvp,sizeof(int)*(size_t) nctg)) != hipSuccess ||
      (e = fga_dmalloc(&ditems,sizeof(gix_item)*items.size())) != hipSuccess ||
      (e = fga_dmalloc(&dcount,sizeof(uint32_t)*FGA_NPREFIX)) != hipSuccess ||
      (e = fga_dmalloc(&dctr,sizeof(unsigned long long)*(1 + 1024 + 4096 + 1))) != hipSuccess ||
      (e = fga_dmalloc(&dpartid,1024)) != hipSuccess)
    { fga_set_error("fga_dgix_build: device allocation failed: %s",hipGetErrorString(e));
      goto done;
    }
  dimg = dimg0 + FGA_IMG_PAD;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(gix_tmap),fga_gix_tmap(),256)) != hipSuccess ||
      (e = hipMemsetAsync(dimg0,0,FGA_IMG_PAD,dev->stream)) != hipSuccess ||
      (e = hipMemsetAsync(dimg + G->bpslen,0,FGA_IMG_PAD + 64,dev->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(dimg,G->bps,(size_t) G->bpslen,hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(dboff,boff.data(),sizeof(int64_t)*boff.size(),hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(dclen,clen.data(),sizeof(int64_t)*clen.size(),hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(dinvp,invp,sizeof(int)*(size_t) nctg,hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
      (e = hipMemcpyAsync(ditems,items.data(),sizeof(gix_item)*items.size(),hipMemcpyHostToDevice,dev->stream)) != hipSuccess ||
      (e = hipMemsetAsync(dcount,0,sizeof(uint32_t)*FGA_NPREFIX,dev->stream)) != hipSuccess ||
      (e = hipMemsetAsync(dctr,0,sizeof(unsigned long long)*(1 + 1024 + 4096 + 1),dev->stream)) != hipSuccess)
    { fga_set_error("fga_dgix_build: upload failed: %s",hipGetErrorString(e));
      goto done;
    }

  tn = fga_wall();
  hipEventRecord(dev->ev0,dev->stream);
  scan_grid = (unsigned) (items.size() < (size_t) dev->ncu*8 ? items.size() : (size_t) dev->ncu*8);
  if (scan_grid < 1) scan_grid = 1;
  { gix_scan_args A;
    A.img = dimg; A.boff = dboff; A.clen = dclen; A.invp = dinvp;
    A.items = ditems; A.nitems = (int) items.size();
    A.postbytes = postbytes; A.contbytes = contbytes;
    A.keys = NULL; A.index = NULL; A.nkeys = dctr; A.count = dcount; A.sbuck = dctr + 1;        // the counting pass
    A.pbeg = (uint32_t) pbeg; A.pend = (uint32_t) pend;
    A.goff = NULL;
    hipLaunchKernelGGL(gix_scan_kernel,dim3(scan_grid),dim3(GNT),0,dev->stream,A);
  }
  if (count_only)
    { if ((e = hipMemcpyAsync(counts_host,dcount,sizeof(uint32_t)*FGA_NPREFIX,hipMemcpyDeviceToHost,dev->stream)) != hipSuccess ||
          (e = hipStreamSynchronize(dev->stream)) != hipSuccess || (e = hipGetLastError()) != hipSuccess)
        { fga_set_error("fga_dgix_prefix_counts: %s",hipGetErrorString(e));
          goto done;
        }
      status = 0;
      goto done;
    }
  { unsigned long long hk[1025];
    if ((e = hipMemcpyAsync(hk,dctr,sizeof(hk),hipMemcpyDeviceToHost,dev->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(dev->stream)) != hipSuccess)
      { fga_set_error("fga_dgix_build: scan kernel failed: %s",hipGetErrorString(e));
        goto done;
      }
    nkeys = (int64_t) hk[0];
    // table parts from the sample histogram, as a bucket -> part id map
    int64_t sb[1024];
    int ksplit[65];
    uint8_t partid[1024];
    for (int b = 0; b < 1024; b++) sb[b] = (int64_t) hk[1+b];
    fga_gix_ksplit(sb,nparts,ksplit);
    { int p = 0;
      for (int b = 0; b < 1024; b++)
        { while (p < nparts && ksplit[p+1] <= b) p += 1;
          partid[b] = (uint8_t) p;
        }
    }
    if ((e = hipMemcpyAsync(dpartid,partid,1024,hipMemcpyHostToDevice,dev->stream)) != hipSuccess)
      { fga_set_error("fga_dgix_build: upload failed: %s",hipGetErrorString(e));
        goto done;
      }
    hipStreamSynchronize(dev->stream);
  }

  fga_note("index build: uploads + syncmer scan (counting pass)",tn); tn = fga_wall();
  D = (fga_dgix *) calloc(1,sizeof(fga_dgix));
  if (D == NULL) { fga_set_error("out of memory"); goto done; }
  D->dev = dev; D->nents = nkeys; D->ebytes = ebytes; D->postbytes = postbytes; D->contbytes = contbytes; D->nctg = nctg;
  // with a host copy wanted the on-disk entry bytes are made (and turned into the view afterwards, as for an uploaded table);
  // otherwise the entries go straight into the view's field arrays: 14 bytes per entry that are never allocated or written
  direct_view = !want_host_copy;
  if (direct_view && (nkeys >= ((int64_t) 5 << 32) - 1024 || postbytes > 4))
    { fga_set_error("genome index of %lld entries, %d position bytes: beyond what a table view holds (5 x 2^32 entries, 4 Gbp contigs)",
                    (long long) nkeys,postbytes);
      goto done;
    }
  if ((!direct_view && (e = fga_dmalloc(&D->table,(size_t) nkeys*ebytes + 64)) != hipSuccess) ||
      (e = fga_dmalloc(&D->index,sizeof(int64_t)*FGA_NPREFIX)) != hipSuccess)
    { fga_set_error("fga_dgix_build: device allocation of the table failed: %s",hipGetErrorString(e));
      goto done;
    }
  if (direct_view && fga_view_alloc(&D->view,nkeys,contbytes,1))
    goto done;
  if (!direct_view)
    hipMemsetAsync(D->table + (size_t) nkeys*ebytes,0,64,dev->stream);
  { unsigned long long *sums = dctr + 1025;
    hipLaunchKernelGGL(gix_index_sums_kernel,dim3(FGA_NPREFIX/ICH),dim3(256),0,dev->stream,dcount,sums);
    hipLaunchKernelGGL(gix_index_scan_kernel,dim3(1),dim3(1024),0,dev->stream,sums,FGA_NPREFIX/ICH,sums);
    hipLaunchKernelGGL(gix_index_write_kernel,dim3(FGA_NPREFIX/ICH),dim3(256),0,dev->stream,dcount,sums,D->index,
                       (unsigned int *) (dctr + 1025 + 4096));
  }
  // the keys, each inside its prefix's panel: the second scan (the counts are what is left of every panel), then the panels
  // put in order tile by tile in LDS; panels beyond a tile's capacity through the LSD radix passes
  if (nkeys > 0)
    { // (a little more than the keys need: the piece is what the comparison's seed buffer -- one seed per entry + 2^20 + a
      //  block per wavefront at human scale, fga_merge.hip -- takes over afterwards, instead of asking the driver for a region)
      buf0 = (uint4 *) fga_dev_acquire(dev,SLOT_SORT0,sizeof(uint4)*(size_t) (nkeys + (1 << 20) + 2*(int64_t) dev->ncu*32*1024 + 4096));
      ntiles = (nkeys + GT_T - 1) / GT_T;
      dtiles = (gix_tile *) fga_dev_acquire(dev,SLOT_TILES,sizeof(gix_tile)*(size_t) (ntiles + 1));
      dover = (gix_over *) fga_dev_acquire(dev,SLOT_MISC,sizeof(gix_over)*(size_t) (nkeys / GT_CAP + 2) + 64);
      if (buf0 == NULL || dtiles == NULL || dover == NULL)
        { fga_set_error("fga_dgix_build: device allocation of the key buffer failed");
          goto done;
        }
      unsigned int *dnover = (unsigned int *) (dover + (nkeys / GT_CAP + 2));
      hipMemsetAsync(dnover,0,sizeof(unsigned int),dev->stream);
      gix_scan_args A;
      A.img = dimg; A.boff = dboff; A.clen = dclen; A.invp = dinvp;
      A.items = ditems; A.nitems = (int) items.size();
      A.postbytes = postbytes; A.contbytes = contbytes;
      A.keys = buf0; A.index = D->index; A.nkeys = dctr; A.count = dcount; A.sbuck = dctr + 1;         // the placing pass
      A.pbeg = (uint32_t) pbeg; A.pend = (uint32_t) pend;
      A.goff = NULL;
      if (wide_pay)
        { if ((e = fga_dmalloc(&dgoff,sizeof(int64_t)*goff.size())) != hipSuccess ||
              (e = hipMemcpyAsync(dgoff,goff.data(),sizeof(int64_t)*goff.size(),hipMemcpyHostToDevice,dev->stream)) != hipSuccess)
            { fga_set_error("fga_dgix_build: device allocation failed: %s",hipGetErrorString(e));
              goto done;
            }
          A.goff = dgoff;
        }
      hipLaunchKernelGGL(gix_scan_kernel,dim3(scan_grid),dim3(GNT),0,dev->stream,A);
      hipLaunchKernelGGL(gix_tiles_kernel,dim3((unsigned) ((ntiles + 1 + 255)/256)),dim3(256),0,dev->stream,D->index,nkeys,ntiles,dtiles);
      hipLaunchKernelGGL(gix_tile_sort_kernel,dim3((unsigned) ntiles),dim3(GT_NT),0,dev->stream,buf0,dtiles,dover,dnover);
      unsigned int nover = 0;
      if ((e = hipMemcpyAsync(&nover,dnover,sizeof(nover),hipMemcpyDeviceToHost,dev->stream)) != hipSuccess ||
          (e = hipStreamSynchronize(dev->stream)) != hipSuccess || (e = hipGetLastError()) != hipSuccess)
        { fga_set_error("fga_dgix_build: placing / panel sort kernels failed: %s",hipGetErrorString(e));
          goto done;
        }
      if (nover > 0)
        { std::vector<gix_over> ov((size_t) nover);
          std::vector<int64_t> ooff((size_t) nover);
          int64_t total = 0;
          if ((e = hipMemcpy(ov.data(),dover,sizeof(gix_over)*(size_t) nover,hipMemcpyDeviceToHost)) != hipSuccess)
            { fga_set_error("fga_dgix_build: download failed: %s",hipGetErrorString(e)); goto done; }
          std::sort(ov.begin(),ov.end(),[](const gix_over &a, const gix_over &b) { return a.start < b.start; });
          for (unsigned int k = 0; k < nover; k++) { ooff[(size_t) k] = total; total += ov[(size_t) k].count; }
          oscr = (uint4 *) fga_dev_acquire(dev,SLOT_SORT1,2*sizeof(uint4)*(size_t) total);
          if (oscr == NULL || (e = fga_dmalloc(&dooff,sizeof(int64_t)*(size_t) nover)) != hipSuccess)
            { fga_set_error("fga_dgix_build: device allocation for %lld keys of %u large panels failed",(long long) total,nover);
              goto done;
            }
          hipMemcpyAsync(dover,ov.data(),sizeof(gix_over)*(size_t) nover,hipMemcpyHostToDevice,dev->stream);
          hipMemcpyAsync(dooff,ooff.data(),sizeof(int64_t)*(size_t) nover,hipMemcpyHostToDevice,dev->stream);
          unsigned cg = (unsigned) ((total + 255)/256 < (int64_t) dev->ncu*16 ? (total + 255)/256 : (int64_t) dev->ncu*16);
          hipLaunchKernelGGL(gix_over_copy_kernel,dim3(cg),dim3(256),0,dev->stream,buf0,oscr,dover,dooff,(int) nover,total,0);
          const int npass = wide_pay ? 16 : (80 + 8*(postbytes+contbytes) + 7) / 8;
          uint4 *osorted = NULL;
          if (fga_radix_sort_u128(dev,oscr,oscr + total,total,128 - 8*npass,8*npass,&osorted))
            goto done;
          hipLaunchKernelGGL(gix_over_copy_kernel,dim3(cg),dim3(256),0,dev->stream,buf0,osorted,dover,dooff,(int) nover,total,1);
          if ((e = hipStreamSynchronize(dev->stream)) != hipSuccess || (e = hipGetLastError()) != hipSuccess)
            { fga_set_error("fga_dgix_build: sort of the large panels failed: %s",hipGetErrorString(e));
              goto done;
            }
          fga_dev_release(dev,SLOT_SORT1,oscr); oscr = NULL;
        }
      sorted = buf0;
    }
  fga_note("index build: placing pass + panel order",tn); tn = fga_wall();
  if (nkeys > 0)
    { gix_entries_args E;
      E.keys = sorted; E.n = nkeys; E.postbytes = postbytes; E.contbytes = contbytes; E.ebytes = ebytes;
      E.table = D->table; E.partid = dpartid;
      E.view = D->view;
      E.moff = NULL; E.mbeg = E.mend = NULL; E.perm = NULL;
      E.goff = wide_pay ? dgoff : NULL; E.ngoff = G->ncontig; E.invp = dinvp;
      if (use_mask)
        { if ((e = fga_dmalloc(&dmoff,sizeof(int64_t)*(size_t) (nctg+1))) != hipSuccess ||
              (e = fga_dmalloc(&dmbeg,sizeof(int64_t)*(size_t) G->nmask)) != hipSuccess ||
              (e = fga_dmalloc(&dmend,sizeof(int64_t)*(size_t) G->nmask)) != hipSuccess ||
              (e = fga_dmalloc(&dperm,sizeof(int)*(size_t) nctg)) != hipSuccess)
            { fga_set_error("fga_dgix_build: device allocation failed: %s",hipGetErrorString(e));
              goto done;
            }
          std::vector<int64_t> mo((size_t) nctg+1);
          for (int c = 0; c <= nctg; c++)
            mo[(size_t) c] = c <= G->ncontig ? G->moff[c] : G->moff[G->ncontig];
          hipMemcpyAsync(dmoff,mo.data(),sizeof(int64_t)*mo.size(),hipMemcpyHostToDevice,dev->stream);
          hipMemcpyAsync(dmbeg,G->mbeg,sizeof(int64_t)*(size_t) G->nmask,hipMemcpyHostToDevice,dev->stream);
          hipMemcpyAsync(dmend,G->mend,sizeof(int64_t)*(size_t) G->nmask,hipMemcpyHostToDevice,dev->stream);
          hipMemcpyAsync(dperm,perm,sizeof(int)*(size_t) nctg,hipMemcpyHostToDevice,dev->stream);
          hipStreamSynchronize(dev->stream);
          E.moff = dmoff; E.mbeg = dmbeg; E.mend = dmend; E.perm = dperm;
        }
      const unsigned egrid = (unsigned) (((nkeys + 255)/256) < (int64_t) (1 << 23) ? ((nkeys + 255)/256) : (int64_t) (1 << 23));
      if (direct_view)
        hipLaunchKernelGGL(gix_entries_kernel<true>,dim3(egrid),dim3(256),0,dev->stream,E);
      else
        hipLaunchKernelGGL(gix_entries_kernel<false>,dim3(egrid),dim3(256),0,dev->stream,E);
    }
  if (direct_view)
    hipLaunchKernelGGL(gix_view_index_kernel,dim3(FGA_NPREFIX/256),dim3(256),0,dev->stream,D->index,D->view.idx);
  hipEventRecord(dev->ev1,dev->stream);
  { unsigned long long hm = 0;
    if ((e = hipMemcpyAsync(&hm,dctr + 1025 + 4096,sizeof(hm),hipMemcpyDeviceToHost,dev->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(dev->stream)) != hipSuccess || (e = hipGetLastError()) != hipSuccess)
      { fga_set_error("fga_dgix_build: kernels failed: %s",hipGetErrorString(e));
        goto done;
      }
    hipEventElapsedTime(&ms,dev->ev0,dev->ev1);
    dev->last_ms[FGA_STAGE_GIX] = ms;
    fga_note("index build: sort + index + entries",tn); tn = fga_wall();
    if (direct_view && fga_view_set_carries(dev,D->index,&D->view))
      goto done;

    X = (fga_gix *) calloc(1,sizeof(fga_gix));
    if (X == NULL) { fga_set_error("out of memory"); goto done; }
    X->kmer = FGA_KMER; X->nparts = nparts; X->postbytes = postbytes; X->contbytes = contbytes; X->ebytes = ebytes;
    X->maxpre = (int64_t) (hm & 0xffffffffu); X->freq = 0; X->nctg = nctg; X->nents = nkeys;
    X->perm = perm; perm = NULL;
    X->partbeg = (int64_t *) malloc(sizeof(int64_t)*(nparts+1));
    if (X->partbeg == NULL)
      { fga_set_error("out of memory"); goto done; }
    // part starts, from the split the device used
    { unsigned long long hk[1025];
      int64_t sb[1024];
      int ksplit[65];
      hipMemcpy(hk,dctr,sizeof(hk),hipMemcpyDeviceToHost);
      for (int b = 0; b < 1024; b++) sb[b] = (int64_t) hk[1+b];
      fga_gix_ksplit(sb,nparts,ksplit);
      for (int p = 0; p <= nparts; p++)
        { X->partbeg[p] = 0;
          if (ksplit[p] > 0 &&
              hipMemcpy(X->partbeg+p,D->index + (((int64_t) ksplit[p] << 14) - 1),sizeof(int64_t),hipMemcpyDeviceToHost) != hipSuccess)
            { fga_set_error("fga_dgix_build: download failed"); goto done; }
        }
    }
    if (want_host_copy)
      { X->index = (int64_t *) malloc(sizeof(int64_t)*FGA_NPREFIX);
        X->table = (uint8_t *) malloc((size_t) nkeys*ebytes + 64);
        if (X->index == NULL || X->table == NULL)
          { fga_set_error("out of memory"); goto done; }
        if ((e = hipMemcpy(X->index,D->index,sizeof(int64_t)*FGA_NPREFIX,hipMemcpyDeviceToHost)) != hipSuccess ||
            (e = hipMemcpy(X->table,D->table,(size_t) nkeys*ebytes + 64,hipMemcpyDeviceToHost)) != hipSuccess)
          { fga_set_error("fga_dgix_build: download failed: %s",hipGetErrorString(e));
            goto done;
          }
      }
    // the merge kernel's view of the table; the on-disk bytes leave the device
    if (!direct_view && fga_dgix_make_view(dev,D,0))
      goto done;
    fga_note("index build: view",tn);
  }
  status = 0;

done:
  if (status == 0 && keep_img != NULL && !count_only)
    *keep_img = dimg0;
  else
    fga_pool_free(dimg0);
  fga_pool_free(dboff); fga_pool_free(dclen); fga_pool_free(dinvp); fga_pool_free(ditems); fga_pool_free(dcount); fga_pool_free(dctr);
  fga_pool_free(dpartid);
  fga_pool_free(dmoff); fga_pool_free(dmbeg); fga_pool_free(dmend); fga_pool_free(dperm); fga_pool_free(dooff); fga_pool_free(dgoff);
  fga_dev_release(dev,SLOT_SORT1,oscr); fga_dev_release(dev,SLOT_TILES,dtiles); fga_dev_release(dev,SLOT_MISC,dover);
  fga_dev_release(dev,SLOT_SORT0,buf0);
  free(perm); free(invp);
  if (status != 0)
    { if (D != NULL) { fga_dgix_free_views(D); fga_pool_free(D->table); fga_pool_free(D->index); free(D); }
      if (X != NULL) fga_gix_close(X);
      return 1;
    }
  if (count_only)
    return 0;
  *dout = D; *xout = X;
  return 0;
}

extern "C" int fga_dgix_build(fga_dev *dev, const fga_gdb *G, int nthreads, int flags,
                              fga_dgix **dout, fga_gix **xout)
{ return dgix_build_impl(dev,G,nthreads,flags,0,FGA_NPREFIX,NULL,dout,xout,NULL); }

// one rank's slice of the table: only the k-mers whose 12-mer prefix lies in [pbeg,pend) ("GPU g uploads only
// its slice of both tables"; the reference's merge threads each read one such range, FastGA.c:2291-2321).  Layout, contig
// order and table parts are those of the whole table; the prefix index counts the slice's entries only.
extern "C" int fga_dgix_build_range(fga_dev *dev, const fga_gdb *G, int nthreads, int flags, int64_t pbeg, int64_t pend,
                                    fga_dgix **dout, fga_gix **xout)
{ if (pbeg < 0 || pend > FGA_NPREFIX || pbeg >= pend)
    { fga_set_error("fga_dgix_build_range: bad prefix range");
      return 1;
    }
  if (flags & FGA_GIX_HOST_COPY)
    { fga_set_error("fga_dgix_build_range: a slice has no host copy");
      return 1;
    }
  return dgix_build_impl(dev,G,nthreads,flags,pbeg,pend,NULL,dout,xout,NULL);
}

// the same two, leaving the genome's padded image on the device for fga_dgenome_adopt (pbeg = 0, pend = 2^24: the whole table)
extern "C" int fga_dgix_build_keep(fga_dev *dev, const fga_gdb *G, int nthreads, int flags, int64_t pbeg, int64_t pend,
                                   fga_dgix **dout, fga_gix **xout, void **image)
{ uint8_t *img = NULL;
  if (image == NULL || pbeg < 0 || pend > FGA_NPREFIX || pbeg >= pend || (flags & FGA_GIX_HOST_COPY))
    { fga_set_error("fga_dgix_build_keep: bad arguments");
      return 1;
    }
  *image = NULL;
  if (dgix_build_impl(dev,G,nthreads,flags,pbeg,pend,NULL,dout,xout,&img))
    return 1;
  *image = img;
  return 0;
}

// entries per 12-mer prefix of the table fga_dgix_build would make (one syncmer scan on the device, nothing is built):
// what the prefix ranges of a sliced run are cut from
extern "C" int fga_dgix_prefix_counts(fga_dev *dev, const fga_gdb *G, int nthreads, uint32_t *counts /* host, 2^24 */)
{ if (counts == NULL)
    { fga_set_error("fga_dgix_prefix_counts: null argument");
      return 1;
    }
  return dgix_build_impl(dev,G,nthreads,0,0,FGA_NPREFIX,counts,NULL,NULL,NULL);
}
