#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4f; mkdir -p $o
{ echo "== sizes ascending, each freed"; fastga_amd/bin/alloc_sizes 2 8 16 24 30 31 32 33 36 40 48 64 38.4 38.4
  echo "== fresh process, 38.4 first"; fastga_amd/bin/alloc_sizes 38.4 38.4 19 19 38.4
  echo "== fresh process, kept: 19 2.4 2.4 9.5 2.4 38.4 38.4 38.4 24"; ALLOC_KEEP=1 fastga_amd/bin/alloc_sizes 19 2.4 2.4 9.5 2.4 38.4 38.4 38.4 24
  echo "== fresh process, kept: 8 x 19.2"; ALLOC_KEEP=1 fastga_amd/bin/alloc_sizes 19.2 19.2 19.2 19.2 19.2 19.2 19.2 19.2
} > $o/alloc.log 2>&1
cat $o/alloc.log
