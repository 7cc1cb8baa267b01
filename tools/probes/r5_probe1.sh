#!/bin/bash
# round 5, first GPU call: the new multi-GPU C entry with virtual ranks, the re-entrancy test, the sharded bench path
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_reentrancy_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5_p1_tests.txt
cat gpurun_out/r5_p1_tests.txt
FGA_BENCH_MULTI_DEVICES=0,0 FGA_BENCH_SHARDED_3G=0 timeout 600 python bench.py --gpus 1 --force-sharded --no-human-scale --steps 3 --warmup 1 --no-cpu > gpurun_out/r5_p1_bench_sharded.json 2> gpurun_out/r5_p1_bench_sharded.err
tail -c 1500 gpurun_out/r5_p1_bench_sharded.json; tail -5 gpurun_out/r5_p1_bench_sharded.err
