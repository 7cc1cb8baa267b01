#!/bin/bash
# round 5, GPU call 6: cutoffs beyond the largest window (window-free merge kernel), 7-byte payloads through the upload
# path and the device builder, hardware queues for the batch block
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5f; mkdir -p $o
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_seed_merge_gpu.py "tests/test_edge_cases_gpu.py::test_seven_byte_payloads_many_small_contigs_and_a_long_one" tests/test_gix_device_gpu.py -x -q -m gpu 2>&1 | tail -25 ) > $o/tests.log 2>&1
tail -12 $o/tests.log
# (GPU_MAX_HW_QUEUES=8 / 16 made the batch run hang until its timeout: the runtime's default of 4 hardware queues stays)
