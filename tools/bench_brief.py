#!/usr/bin/env python3
"""Run bench.py with the given arguments and print a one-screen digest of its JSON line (for gpurun tails)."""
import json
import subprocess
import sys

r = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
for ln in r.stderr.splitlines():
    if "profile" in ln or "rror" in ln:
        print(ln[:400])
for ln in r.stdout.splitlines():
    if ln.startswith("{"):
        d = json.loads(ln)
        c = d["config"]
        print("ms/step %.2f  value %.4f %s  roofline %.3f (%.0f GB/s)" % (d["ms_per_step"], d["value"], d["unit"],
              d["roofline"]["frac"], d["roofline"]["achieved"]))
        print("stage_ms", c["stage_ms"])
        print("kernel_ms", c["kernel_ms"], "seeds", c["seeds"], "hits", c["hits"], "alns", c["alignments"], "records", c["records"])
        if "cpu_baseline" in d:
            b = d["cpu_baseline"]
            print("cpu", b.get("value"), b.get("kind"), "identical_1aln", b.get("identical_1aln"), "strict", b.get("identical_1aln_strict"))
