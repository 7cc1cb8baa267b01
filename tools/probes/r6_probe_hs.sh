#!/bin/bash
# round 6: the 3 Gbp legs of bench.py (configs[3] / configs[4] on one GPU), reference leg off
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r6h; mkdir -p $o
export TMPDIR=/tmp
FGA_BENCH_REF_3G=${REF3G:-0} FGA_TIMING=${TIMING:-0} timeout 1500 python bench.py --steps 5 --warmup 2 --no-cpu --batch 0 > $o/hs.json 2> $o/hs.err
python - $o/hs.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "kernel_ms", d["config"]["kernel_ms"])
print("cold", d.get("cold", {}).get("seconds"))
for k in ("human_scale", "human_scale_10pct"):
    h = d.get(k)
    if not h: continue
    if "error" in h: print(k, h); continue
    print(k, "seconds", h["seconds"], "first", h["first_run_seconds"], "kernel_ms", h["kernel_ms"], "stage_s", h["stage_s"], "merge frac", round(h["roofline"]["frac"], 3))
    print("   sort", {a: h["sort"][a] for a in ("passes", "kernel_ms", "frac")}, "cold", h["cold"]["seconds"], h["cold"].get("of_which_driver_alloc_s"), "open", h["upload_and_2_index_builds_s"], "eq", h.get("counts_equal_reference"), "ref", h.get("reference"))
    if "projected_8gpu" in h:
        p = h["projected_8gpu"]
        print("   proj", {a: p[a] for a in ("seconds", "phase1_s_max", "phase2_s_max", "finish_s", "streamed_finish", "appends_after_slowest_rank_s", "part_imbalance_extend", "part_imbalance_seeds", "records_equal_one_gpu_run", "first_run_of_a_session")})
PY
grep -i "timing\|error\|fga" $o/hs.err | tail -${TLINES:-5}
