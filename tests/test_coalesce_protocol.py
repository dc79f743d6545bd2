"""The class API's cross-thread combiner (csrc/coalesce.hpp, host/coalesce.cpp) WITHOUT a GPU: its protocol -- one queue and service
thread per kind of call, short spins and futex sleeps, wake-ups, requests queued ahead of their caller (the prefetch of a row's
other pricers) and taken later or drained when the thread's matrix changes or the thread ends, arena slots -- driven by many host
threads against a stand-in for the device that answers a round after ~50 us (mcg_debug_coalesce_selftest, include/mcgpu_debug.h).
What the GPU tests add (tests/test_gpu_round6.py) is the device side: the row kernels' answers and the 128-thread driver."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
import montecarlooptionspricer_amd as mc
L = mc.load_library()
wrong = C.c_int(-1)
rc = L.mcg_debug_coalesce_selftest(int(sys.argv[1]), int(sys.argv[2]), C.byref(wrong))
print(json.dumps({"rc": rc, "wrong": wrong.value, "stats": mc.stats()}))
""" % ROOT


@pytest.mark.parametrize("threads,calls", [(1, 400), (8, 600), (96, 250)])
def test_combiner_protocol_answers_every_call_of_every_thread(threads, calls):
    """Every call gets ITS answer (a function of its own arguments), whatever shares its round; nothing dead-locks (the child
    process is timed out); requests queued ahead are answered too -- taken, or drained --; the process ends cleanly with its five
    service threads joined.  One thread alone: rounds of at most two calls (its own call meeting its own prefetched request)."""
    p = subprocess.run([sys.executable, "-c", CHILD, str(threads), str(calls)], capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["rc"] == 0 and j["wrong"] == 0, j
    s = j["stats"]
    n_sync = threads * calls
    n_ahead = threads * 2 * len([c for c in range(calls) if c % 3 == 0])
    assert s["coalesced_calls"] == n_sync + n_ahead and s["coalesced_prefetched"] == n_ahead
    assert s["coalesced_prefetch_hits"] == threads * len([c for c in range(calls) if c % 6 == 0])
    assert s["coalesced_rounds"] <= s["coalesced_calls"]
    if threads == 1:
        assert s["coalesced_peak_calls_per_round"] <= 2
    else:
        assert s["coalesced_peak_calls_per_round"] >= 2 and s["coalesced_rounds"] < s["coalesced_calls"]


def test_combiner_protocol_is_clean_under_thread_sanitizer(tmp_path):
    """host/coalesce.cpp compiled by itself with -fsanitize=thread (g++; tests/cpp/coalesce_tsan_main.cpp supplies the handful of
    symbols it takes from the rest of the library) and driven by the same self-test: 24 threads x 300 calls -- no data race, no wrong
    answer.  (The GPU box has no sanitizer runs: this is the CPU build the sanitizers are allowed on.)"""
    exe = str(tmp_path / "coalesce_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + ROOT,
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "coalesce_tsan_main.cpp"),
           os.path.join(ROOT, "montecarlooptionspricer_amd", "host", "coalesce.cpp"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
           "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    p = subprocess.run([exe, "24", "300"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
    assert p.returncode == 0 and "WARNING: ThreadSanitizer" not in p.stderr and p.stdout.startswith("wrong 0 "), p.stdout[-500:] + p.stderr[-3000:]
