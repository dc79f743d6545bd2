"""The C++ boundary end to end: tests/cpp/dropin_driver.cpp is written against the reference's
public class API only, built with plain g++ against include/ and libmcgpu.so, and run the way the
reference's driver runs its pricers (per-row objects inside an OpenMP parallel-for,
src/core/PredictionGen.cpp:542-570).  Exercises re-entrancy (one lazily created ctx per host thread)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_dropin_driver_openmp():
    subprocess.run(["make", "cpp"], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, OMP_NUM_THREADS="6")
    res = subprocess.run([os.path.join(ROOT, "build", "dropin_driver"), "24"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "OK rows=24"
    rows = [l for l in lines if l.startswith("row ")]
    assert len(rows) == 24 and not any("EXCEPTION" in l for l in rows)
    # fixed seed => rows with the same step count are identical whatever thread priced them
    by_steps = {}
    for l in rows:
        tok = l.split()
        by_steps.setdefault(tok[3], set()).add((tok[5], tok[7]))
    assert all(len(v) == 1 for v in by_steps.values()), by_steps
