#!/bin/bash
# round 4, GPU call 2: the index build without global radix passes (placing pass + LDS panel order)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4b; mkdir -p $o
( timeout 1500 python -m pytest tests/test_gix_device_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py tests/test_end_to_end_gpu.py "tests/test_full_size_gpu.py::test_config2_100mbp_pair_is_identical_to_the_reference" -x -q -m gpu 2>&1 | tail -25 ) > $o/tests.log 2>&1
tail -25 $o/tests.log
FGA_TIMING=1 timeout 600 python tools/config4_check.py --mbp 3000 --div 0.01 > $o/c4.log 2>&1
grep -v "^\[fga timing\] \(pool\|region\)" $o/c4.log | grep -i "timing\|upload\|session_run\|stages\|digest\|error\|Traceback" | tail -40
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from fastga_amd import workload
import os
g = json.load(open("tests/golden/config4_3000m_digest.json"))
last = [ln for ln in open("gpurun_out/r4b/c4.log") if ln.startswith("{")]
if last:
    r = json.loads(last[-1])
    d = r.get("ours_digest", {})
    print("3 Gbp digest == golden:", all(d.get(k) == g[k] for k in ("records", "header_md5", "records_sum128", "order_md5")))
PY
