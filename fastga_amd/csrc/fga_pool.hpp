// fga_pool.hpp -- the piece bookkeeping of the device-memory pool (fga_device.hip), free of HIP: the regions come from a
// backend (hipMalloc / hipFree there, malloc / free in tests/pool_host_test.cpp, which runs this file on the host).
// A request takes the smallest free piece that holds it (split, the rest stays free); a release merges the piece with
// its free neighbours; a region goes back to the backend only by trim() (regions that are entirely free) -- which take()
// calls itself, once, when the backend cannot provide a new region.  No locking here: the caller serialises.
#pragma once
#include <stddef.h>
#include <vector>

struct fga_pool_piece  { char *ptr; size_t bytes; int region; bool busy; };
struct fga_pool_region { char *base; size_t bytes; };
struct fga_pool_backend
  { int  (*alloc)(void **out, size_t bytes);      // 0: done
    void (*release)(void *ptr);
  };

struct fga_pool_core
  { std::vector<fga_pool_piece>  pieces;          // by (region, address): the pieces of a region tile it
    std::vector<fga_pool_region> regions;         // slot r stays r while the region lives (base NULL: slot free)

    void trim(const fga_pool_backend &B)
    { for (size_t k = 0; k < pieces.size(); )
        { const fga_pool_piece &q = pieces[k];
          if (!q.busy && q.ptr == regions[(size_t) q.region].base && q.bytes == regions[(size_t) q.region].bytes)
            { B.release(q.ptr);
              regions[(size_t) q.region].base = NULL; regions[(size_t) q.region].bytes = 0;
              pieces.erase(pieces.begin() + (long) k);
            }
          else
            k += 1;
        }
    }

    // `need` bytes (the caller has rounded them to its granule); *fresh = a new region was taken from the backend
    void *take(size_t need, const fga_pool_backend &B, bool *fresh)
    { *fresh = false;
      long best = -1;
      for (size_t k = 0; k < pieces.size(); k++)
        if (!pieces[k].busy && pieces[k].bytes >= need && (best < 0 || pieces[k].bytes < pieces[(size_t) best].bytes))
          best = (long) k;
      if (best >= 0)
        { char *p = pieces[(size_t) best].ptr;
          const size_t rest = pieces[(size_t) best].bytes - need;
          pieces[(size_t) best].busy = true; pieces[(size_t) best].bytes = need;
          if (rest > 0)
            { const fga_pool_piece r = { p + need, rest, pieces[(size_t) best].region, false };
              pieces.insert(pieces.begin() + best + 1,r);
            }
          return p;
        }
      void *p = NULL;
      if (B.alloc(&p,need) != 0)
        { trim(B);                                // regions nobody uses go back, then once more
          if (B.alloc(&p,need) != 0)
            return NULL;
        }
      *fresh = true;
      int r = -1;
      for (size_t k = 0; k < regions.size(); k++)
        if (regions[k].base == NULL) { r = (int) k; break; }
      if (r < 0) { r = (int) regions.size(); regions.push_back(fga_pool_region()); }
      regions[(size_t) r].base = (char *) p; regions[(size_t) r].bytes = need;
      size_t at = pieces.size();                  // behind the pieces of the regions before it
      for (size_t k = 0; k < pieces.size(); k++)
        if (pieces[k].region > r) { at = k; break; }
      const fga_pool_piece q = { (char *) p, need, r, true };
      pieces.insert(pieces.begin() + (long) at,q);
      return p;
    }

    bool holds(const void *ptr) const
    { for (const fga_pool_piece &q : pieces)
        if (q.ptr == (const char *) ptr && q.busy)
          return true;
      return false;
    }

    bool give(void *ptr)                          // false: not a busy piece of this pool
    { size_t k = 0;
      for (; k < pieces.size(); k++)
        if (pieces[k].ptr == (char *) ptr && pieces[k].busy)
          break;
      if (k == pieces.size())
        return false;
      pieces[k].busy = false;
      if (k+1 < pieces.size() && !pieces[k+1].busy && pieces[k+1].region == pieces[k].region)
        { pieces[k].bytes += pieces[k+1].bytes;
          pieces.erase(pieces.begin() + (long) k + 1);
        }
      if (k > 0 && !pieces[k-1].busy && pieces[k-1].region == pieces[k].region)
        { pieces[k-1].bytes += pieces[k].bytes;
          pieces.erase(pieces.begin() + (long) k);
        }
      return true;
    }

    void idle(size_t *total, size_t *largest) const
    { *total = *largest = 0;
      for (const fga_pool_piece &q : pieces)
        if (!q.busy)
          { *total += q.bytes;
            if (q.bytes > *largest) *largest = q.bytes;
          }
    }
  };
