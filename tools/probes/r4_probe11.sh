#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
( time timeout 1200 python -m pytest tests/test_seed_merge_gpu.py tests/test_end_to_end_gpu.py -x -q -m gpu --durations=5 2>&1 | tail -14 )
