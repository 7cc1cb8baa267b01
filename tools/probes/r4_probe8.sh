#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4h; mkdir -p $o
echo skip smoke
FGA_BENCH_SHARDED_3G=force FGA_BENCH_SHARDED_MBP=300 timeout 600 python bench.py --force-sharded --steps 2 --warmup 1 --no-cpu --no-cold --mbp 20 > $o/bench_sharded.json 2> $o/bench_sharded.err; echo "bench rc $?"
tail -3 $o/bench_sharded.err | cut -c1-300
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4h/bench_sharded.json").read().splitlines()[0])
print({k: j[k] for k in ("value", "ms_per_step", "n_gpus")})
print(json.dumps(j.get("human_scale_sharded"))[:1500])
PY
