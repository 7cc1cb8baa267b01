/* unlink_shim.c -- TEST INFRASTRUCTURE (oracle/): LD_PRELOAD shim used only when running the real
 * reference FastGA from oracle/_ref to capture its intermediate seed streams.
 *
 * FastGA unlinks its `_pair.<pid>.<k>.{N,C}` seed files right after creating them (reference
 * FastGA.c:5119-5132).  With this shim preloaded, unlink() of a path containing "_pair." is ignored, so
 * the exact seed records phase 1 wrote survive in the -P directory for comparison.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <string.h>
#include <unistd.h>

int unlink(const char *path)
{ static int (*real)(const char *) = NULL;
  if (real == NULL)
    real = (int (*)(const char *)) dlsym(RTLD_NEXT,"unlink");
  if (path != NULL && strstr(path,"_pair.") != NULL)
    return 0;
  return real(path);
}
