#!/usr/bin/env python3
"""tools/isa_budget.py -- instruction counts of one device function of fga_extend.hip, per basic block and class.

  python tools/isa_budget.py [--func ext_mid12ext_episode1ILi1] [--blocks] [--dump] [-D...]

Compiles fastga_amd/csrc/fga_extend.hip for gfx950 with the Makefile's per-file flags (device code only, -S), cuts the named
function out of the assembly and counts VALU / SALU / LDS / VMEM / SMEM / branch / wait instructions per basic block.  The
common path of a wave step is a handful of these blocks (profiles/r06_extend_step_budget.txt names them); the per-category
budget there is this output with the instructions of the long blocks assigned to what they compute.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fastga_amd", "csrc", "fga_extend.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I" + os.path.join(ROOT, "include"),
         "-I" + os.path.join(ROOT, "fastga_amd", "csrc"), "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-slp-vectorize",
         "--cuda-device-only", "-S"]


def klass(op):
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def assemble(defs):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + defs + ["-o", out, SRC], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def function_lines(path, key):
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if start is None and re.match(r"^_Z\w*:", ln) and key in ln:
            start = i
        elif start is not None and ln.startswith(".Lfunc_end"):
            return lines[start:i]
    raise SystemExit(f"no function matching {key!r}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--func", default="ext_mid12ext_episode1ILi1")
    ap.add_argument("--blocks", action="store_true", help="per basic block (default: whole function)")
    ap.add_argument("--dump", action="store_true", help="print the function's instructions with their class")
    ap.add_argument("--asm", help="use this assembly file instead of compiling")
    args, defs = ap.parse_known_args()
    path = args.asm or assemble(defs)
    body = function_lines(path, args.func)
    order = ["valu", "salu", "lds", "vmem", "smem", "branch", "wait"]
    blocks, cur = [], ["entry", dict.fromkeys(order, 0), []]
    for ln in body[1:]:
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            blocks.append(cur)
            cur = [m.group(1), dict.fromkeys(order, 0), []]
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        k = klass(op)
        if k in cur[1]:
            cur[1][k] += 1
        cur[2].append((k, t.split(";")[0].rstrip()))
    blocks.append(cur)
    if args.dump:
        for name, _, ins in blocks:
            print(f"{name}:")
            for k, t in ins:
                print(f"  {k:6s} {t}")
        return
    tot = dict.fromkeys(order, 0)
    print(f"# {args.func}: " + " ".join(f"{k:>6s}" for k in order))
    for name, cnt, _ in blocks:
        for k in order:
            tot[k] += cnt[k]
        if args.blocks:
            print(f"{name:12s} " + " ".join(f"{cnt[k]:6d}" for k in order))
    print(f"{'total':12s} " + " ".join(f"{tot[k]:6d}" for k in order))
    if not args.asm:
        os.unlink(path)


if __name__ == "__main__":
    sys.exit(main())
