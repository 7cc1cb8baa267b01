#!/bin/bash
# round 5, GPU call 5: the whole GPU suite over the pool whose releases wait for the owner's stream only, then the bench
# line's batch block (eight comparisons in flight)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5e; mkdir -p $o
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $o/tests.log 2>&1
tail -6 $o/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-human-scale --no-cpu --no-cold > $o/bench.json 2> $o/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5e/bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 4), "ms/step", round(d["ms_per_step"], 2), "kernel_ms", d["config"]["kernel_ms"], "stage", d["config"]["stage_ms"])
print("batch", d.get("batch"))
PY
tail -3 $o/bench.err
for k in 4 16; do timeout 300 python bench.py --steps 6 --warmup 2 --no-human-scale --no-cpu --no-cold --batch $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['batch']; print('batch', b['comparisons_in_flight'], round(b['value'],3), 'Gbp-pair/s', b['ms_per_comparison_amortised'], 'ms each', round(b['vs_one_at_a_time'],2), 'x')"; done
