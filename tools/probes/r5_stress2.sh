#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5s3; mkdir -p $o
export TMPDIR=/tmp
timeout 400 python tools/multi_stress.py 40 > $o/plain.log 2>&1; tail -12 $o/plain.log | cut -c1-400
echo "== FGA_POOL_DEVICE_SYNC=1"
FGA_POOL_DEVICE_SYNC=1 timeout 400 python tools/multi_stress.py 40 > $o/devsync.log 2>&1; tail -6 $o/devsync.log | cut -c1-400
