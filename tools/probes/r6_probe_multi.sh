#!/bin/bash
# round 6: bench.py's N > 1 path on one GPU (ranks sharing GPU 0), its legs at a reduced size, and the N = 1 line
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r6m; mkdir -p $o
export TMPDIR=/tmp
FGA_BENCH_SHARDED_3G=force FGA_BENCH_SHARDED_MBP=300 timeout 900 python bench.py --devices 0,0,0,0 --steps 4 --warmup 1 > $o/multi4.json 2> $o/multi4.err
python - $o/multi4.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 2), "value", round(d["value"], 3), "n_gpus", d["n_gpus"], "ranks", d["config"]["ranks"])
print("per_rank", d["config"].get("per_rank"))
print("parity", d.get("parity")); print("cold", d.get("c_abi_multi_cold"))
hs = d.get("human_scale_sharded"); print("sharded", {k: hs[k] for k in hs if k not in ("workload",)} if hs else None)
PY
tail -3 $o/multi4.err
timeout 600 python tools/bench_brief.py --steps 5 --warmup 2 --no-human-scale --no-cold --batch 0 2>&1 | tail -6
