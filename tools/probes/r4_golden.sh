#!/bin/bash
# round 4: golden digests with the order-sensitive lines_md5 (the reference at 3 Gbp and 1 Gbp on the GPU box's host cores)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/golden; mkdir -p $o
timeout 1500 python tests/golden/make_golden_config4.py --threads 32 --workdir /dev/shm/fga_golden --outdir $o > $o/c45.log 2>&1
grep -i "reference\|identical\|ours ==\|session_run\|Traceback\|Error" $o/c45.log | cut -c1-260
timeout 900 python tests/golden/make_golden_config3.py --threads 8 --device-index --workdir /dev/shm/fga_golden_c3 --out $o/config3_1000m_digest.json > $o/c3.log 2>&1
rm -rf /dev/shm/fga_golden_c3 /dev/shm/fga_golden
tail -3 $o/c3.log | cut -c1-600
