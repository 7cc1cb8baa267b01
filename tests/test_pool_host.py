"""CPU: the piece bookkeeping of the device-memory pool (fastga_amd/csrc/fga_pool.hpp, what fga_device.hip runs over
hipMalloc) compiled for the host over malloc and driven with random requests / releases against a backend that refuses
beyond a cap; tests/native/pool_host_test.cpp checks the invariants after every step."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_bookkeeping_on_the_host(tmp_path):
    exe = str(tmp_path / "pool_host_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "fastga_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "pool_host_test.cpp"), "-o", exe], check=True)
    for seed in (1, 2, 3, 4):
        r = subprocess.run([exe, str(seed), "12000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        assert r.stdout.startswith("ok seed")
