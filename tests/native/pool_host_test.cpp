// tests/native/pool_host_test.cpp -- the piece bookkeeping of the device-memory pool (fastga_amd/csrc/fga_pool.hpp) run on
// the host over malloc: random requests and releases, with a backend that refuses beyond a cap (so that take() has to
// give idle regions back), checked after every step: busy pieces never overlap, the pieces of a region tile it, free
// neighbours are merged, a released region is gone, and everything released means every region is one free piece.
#include "fga_pool.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <map>

static size_t backend_bytes = 0, backend_cap = 0;
static std::map<void *,size_t> backend_live;

static int t_alloc(void **out, size_t bytes)
{ if (backend_bytes + bytes > backend_cap) { *out = NULL; return 1; }
  *out = malloc(bytes);
  if (*out == NULL) return 1;
  backend_live[*out] = bytes; backend_bytes += bytes;
  return 0;
}
static void t_release(void *p)
{ auto it = backend_live.find(p);
  if (it == backend_live.end()) { fprintf(stderr,"release of an unknown region\n"); exit(2); }
  backend_bytes -= it->second; backend_live.erase(it);
  free(p);
}
static const fga_pool_backend B = { t_alloc, t_release };

static void check(const fga_pool_core &P, const std::map<char *,size_t> &mine)
{ // regions known to the pool = regions live in the backend
  size_t nreg = 0;
  for (const fga_pool_region &r : P.regions)
    if (r.base != NULL)
      { nreg += 1;
        if (backend_live.count(r.base) == 0 || backend_live[r.base] != r.bytes) { fprintf(stderr,"region mismatch\n"); exit(3); }
      }
  if (nreg != backend_live.size()) { fprintf(stderr,"region count %zu vs %zu\n",nreg,backend_live.size()); exit(3); }
  // the pieces of a region tile it, in address order; no two free neighbours
  for (size_t k = 0; k < P.pieces.size(); k++)
    { const fga_pool_piece &q = P.pieces[k];
      const fga_pool_region &r = P.regions[(size_t) q.region];
      const bool first = (k == 0 || P.pieces[k-1].region != q.region);
      const bool last  = (k+1 == P.pieces.size() || P.pieces[k+1].region != q.region);
      if (first && q.ptr != r.base) { fprintf(stderr,"first piece not at the region's base\n"); exit(4); }
      if (!first && P.pieces[k-1].ptr + P.pieces[k-1].bytes != q.ptr) { fprintf(stderr,"gap or overlap\n"); exit(4); }
      if (last && q.ptr + q.bytes != r.base + r.bytes) { fprintf(stderr,"last piece does not end the region\n"); exit(4); }
      if (!first && !q.busy && !P.pieces[k-1].busy) { fprintf(stderr,"free neighbours not merged\n"); exit(4); }
      if (k > 0 && P.pieces[k-1].region > q.region) { fprintf(stderr,"pieces out of region order\n"); exit(4); }
    }
  // what the test holds is exactly the busy pieces
  size_t nbusy = 0;
  for (const fga_pool_piece &q : P.pieces)
    if (q.busy)
      { nbusy += 1;
        auto it = mine.find(q.ptr);
        if (it == mine.end() || it->second != q.bytes) { fprintf(stderr,"busy piece unknown to the test\n"); exit(5); }
      }
  if (nbusy != mine.size()) { fprintf(stderr,"busy count\n"); exit(5); }
}

int main(int argc, char **argv)
{ const unsigned seed = argc > 1 ? (unsigned) atoi(argv[1]) : 1u;
  const int steps = argc > 2 ? atoi(argv[2]) : 20000;
  srand(seed);
  const size_t G = 4096;                          // the granule of this test
  backend_cap = 4096*G;
  fga_pool_core P;
  std::map<char *,size_t> mine;
  size_t fresh_count = 0, refused = 0, reused = 0;
  for (int s = 0; s < steps; s++)
    { const bool doalloc = mine.empty() || (rand() % 100) < 55;
      if (doalloc)
        { const size_t need = G * (size_t) (1 + rand() % ((rand() % 8 == 0) ? 600 : 40));
          bool fresh = false;
          char *p = (char *) P.take(need,B,&fresh);
          if (p == NULL) refused += 1;
          else
            { if (mine.count(p)) { fprintf(stderr,"a busy piece handed out twice\n"); return 6; }
              for (size_t x = 0; x < need; x += G) p[x] = (char) s;        // the memory is real
              mine[p] = need;
              if (fresh) fresh_count += 1; else reused += 1;
            }
        }
      else
        { auto it = mine.begin();
          std::advance(it,(long) (rand() % (int) mine.size()));
          if (!P.holds(it->first) || !P.give(it->first)) { fprintf(stderr,"give refused a busy piece\n"); return 7; }
          if (P.give(it->first)) { fprintf(stderr,"a piece released twice\n"); return 7; }
          mine.erase(it);
        }
      check(P,mine);
      if (s % 997 == 0)
        { P.trim(B); check(P,mine); }
    }
  if (P.give((void *) &backend_cap)) { fprintf(stderr,"a foreign pointer accepted\n"); return 8; }
  while (!mine.empty())
    { P.give(mine.begin()->first); mine.erase(mine.begin()); }
  check(P,mine);
  for (size_t k = 0; k < P.pieces.size(); k++)
    if (P.pieces[k].busy || (k > 0 && P.pieces[k-1].region == P.pieces[k].region)) { fprintf(stderr,"not one free piece per region\n"); return 9; }
  size_t tot, big;
  P.idle(&tot,&big);
  if (tot != backend_bytes) { fprintf(stderr,"idle bytes\n"); return 9; }
  P.trim(B);
  if (!backend_live.empty() || backend_bytes != 0 || !P.pieces.empty()) { fprintf(stderr,"trim left regions\n"); return 10; }
  printf("ok seed %u: %d steps, %zu regions taken, %zu requests served from free pieces, %zu refused at the cap\n",seed,steps,fresh_count,reused,refused);
  return 0;
}
