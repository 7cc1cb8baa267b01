#!/bin/bash
# round 5, GPU call 9: the .1aln writer with every formatter job writing its own stretch of the file (pwrite), 3 Gbp timeline
# (cold + warm) and the 3 Gbp digests against the reference's goldens
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5i; mkdir -p $o
export TMPDIR=/tmp
FGA_TIMING=1 timeout 900 python tools/config4_check.py --mbp 3000 --div 0.01 --runs 3 > $o/c4.log 2>&1
grep -v "pool\|region\|hipMalloc" $o/c4.log | grep -i "comparison\|finish\|filters\|session_run\|stages\|digest" | tail -30
python - <<'PY'
import json
g = json.load(open("tests/golden/config4_3000m_digest.json"))
last = [ln for ln in open("gpurun_out/r5i/c4.log") if ln.startswith("{")]
if last:
    d = json.loads(last[-1]).get("ours_digest", {})
    print("3 Gbp digest == golden (incl. lines_md5):", all(d.get(k) == g[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5")))
PY
( timeout 600 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $o/tests.log 2>&1; tail -3 $o/tests.log
