#!/bin/bash
# device Gap_Improver: parity tests, then the 1.02 M-alignment shape with the phase split of -pafx, then the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python -m pytest tests/test_trace_gpu.py tests/test_extend_gpu.py tests/test_end_to_end_gpu.py -x -q 2>&1 | tail -5
FGA_TRACE_TIMING=1 FGA_PAF_TIMING=1 timeout 300 python tools/scale_check.py --mbp 150 --self --pafx 2>&1 | grep -v "^synth\|^GDB\|^load" | cut -c1-250
FGA_TRACE_TIMING=1 FGA_PAF_TIMING=1 timeout 300 python tools/paf_bench.py 2>&1 | tail -12 | cut -c1-250
