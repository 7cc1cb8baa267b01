#!/bin/bash
# after the fills moved to the contexts' streams: fga_run_multi with virtual ranks again and again, then the suites around it
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5s5; mkdir -p $o
export TMPDIR=/tmp
timeout 600 python tools/multi_stress.py 150 > $o/stress.log 2>&1; echo "$(tail -1 $o/stress.log)"; grep "^iteration" $o/stress.log | head -3 | cut -c1-200
( timeout 900 python -m pytest tests/test_parts_gpu.py tests/test_multi_gpu.py tests/test_reentrancy_gpu.py tests/test_end_to_end_gpu.py tests/test_shims_gpu.py tests/test_edge_cases_gpu.py -q -m gpu 2>&1 | tail -4 ) > $o/t.log 2>&1; tail -1 $o/t.log
