#!/bin/bash
# round 5, GPU call 4: the Compute_Trace_PTS / Gap_Improver shims against libalign_ref.so, seed-merge tests with the cached
# range cuts, and the bench line with the batch block (N = 1, no 3 Gbp legs)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5d; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_shims_gpu.py tests/test_seed_merge_gpu.py tests/test_parts_gpu.py -x -q -m gpu 2>&1 | tail -12 ) > $o/tests.log 2>&1
tail -6 $o/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-human-scale > $o/bench.json 2> $o/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5d/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 4), "ms/step", round(d["ms_per_step"], 2), "kernel_ms", d["config"]["kernel_ms"])
print("roofline", {k: d["roofline"][k] for k in ("frac", "kernel_ms", "walk_kernel_ms", "walk_kernel_frac")})
print("batch", d.get("batch"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "identical_1aln_strict", "cores")})
PY
tail -3 $o/bench.err
FGA_MERGE_CUT_CACHE=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-human-scale --no-cpu --no-cold --batch 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no cut cache:', {k: d['roofline'][k] for k in ('frac','kernel_ms','walk_kernel_ms')})"
