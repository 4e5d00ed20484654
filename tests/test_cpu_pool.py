"""The persistent host threads of the library and the driver (csrc/pgm_pool.h): every index exactly once, exceptions reported,
nested and back-to-back sections (late risers of one section must not touch the next one's counters)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pool_stress(tmp_path):
    exe = str(tmp_path / "pool_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "prographmsa_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tests", "native", "pool_test.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
