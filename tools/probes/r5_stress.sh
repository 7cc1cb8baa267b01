#!/bin/bash
# round 5: repeat the two flows that failed once in tools/r5_final.sh
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5s2; mkdir -p $o
export TMPDIR=/tmp
( timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 )
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --force-sharded --steps 5 --warmup 2 --no-human-scale > $o/bs_$i.json 2> $o/bs_$i.err
  echo "sharded run $i: rc=$? bytes=$(wc -c < $o/bs_$i.json) $(grep -v 'amdgpu.ids\|socket.cpp' $o/bs_$i.err | tail -3 | cut -c1-300)"
done
for i in 1 2 3; do
  ( timeout 600 python -m pytest tests/test_parts_gpu.py tests/test_multi_gpu.py tests/test_reentrancy_gpu.py -q -m gpu 2>&1 | tail -25 ) > $o/t_$i.log 2>&1
  echo "tests run $i: $(tail -1 $o/t_$i.log)"; grep -B2 -A12 "Error\|assert" $o/t_$i.log | head -40 | cut -c1-250
done
