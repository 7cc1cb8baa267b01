#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4suite; mkdir -p $o
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 2>&1 | tail -40 ) > $o/suite.log 2>&1
tail -45 $o/suite.log
