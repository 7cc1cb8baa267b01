#!/bin/bash
# round 5, GPU call 10: the .1aln writer's data part through a shared mapping (3 Gbp, three comparisons of one session)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5j; mkdir -p $o
export TMPDIR=/tmp
FGA_TIMING=1 timeout 900 python tools/config4_check.py --mbp 3000 --div 0.01 --runs 3 > $o/c4.log 2>&1
grep -v "pool\|region\|hipMalloc" $o/c4.log | grep -i "comparison\|finish\|filters\|session_run\|stages" | tail -30
python - <<'PY'
import json
g = json.load(open("tests/golden/config4_3000m_digest.json"))
last = [ln for ln in open("gpurun_out/r5j/c4.log") if ln.startswith("{")]
if last:
    d = json.loads(last[-1]).get("ours_digest", {})
    print("3 Gbp digest == golden (incl. lines_md5):", all(d.get(k) == g[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5")))
PY
