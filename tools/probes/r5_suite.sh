#!/bin/bash
# the whole GPU suite (what the driver runs at round end), tail into gpurun_out/r5_suite.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 ) > gpurun_out/r5_suite.txt 2>&1
tail -32 gpurun_out/r5_suite.txt
