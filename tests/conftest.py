"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """After a GPU run: the library's process-wide event counters (mcg_stats) -- one-launch sweeps that timed out and fell
    back, per-date sweeps that faulted, barrier failures, peer-memory mailboxes refused -- so that a slow or odd run can be
    read from its own log instead of being re-run.  Child processes print theirs into their results."""
    try:
        import montecarlooptionspricer_amd as mc
        s = mc.stats()
    except Exception:   # noqa: BLE001 -- library not built / not loadable: nothing to report
        return
    if any(s.values()):
        terminalreporter.write_sep("-", "libmcgpu event counters of this process (mcg_stats)")
        terminalreporter.write_line(", ".join(f"{k}={v}" for k, v in s.items()))
