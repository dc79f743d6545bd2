"""The node segment's host protocol (montecarlooptionspricer_amd/csrc/comm_shm.cpp) without a GPU: joining, the
stale-segment check, the barrier and its poisoning -- through the library's host-only hooks (mcg_debug_shm_*), one
thread per rank (ctypes releases the GIL for the duration of a call)."""
import ctypes as C
import os
import struct
import threading
import time

import pytest

from montecarlooptionspricer_amd import _native as N

MAGIC = 0x4D434755   # comm_shm.cpp: SHM_MAGIC (layout of round 4)
SEG_BYTES = 16 << 20     # >= sizeof(ShmHeader) + the mailbox


@pytest.fixture(scope="module")
def L():
    return N.load_library()


def _attach(L, name, world, rank, timeout, out, key):
    h = C.c_void_p()
    rc = L.mcg_debug_shm_attach(name.encode(), world, rank, timeout, C.byref(h))
    out[key] = (rc, h, L.mcg_last_error().decode() if rc else "")


def _plant_stale_segment(name, world, attached):
    """What a job that crashed after (or during) set-up leaves behind: a segment of the right size under the same name,
    valid magic, the same rank count."""
    path = "/dev/shm" + name
    with open(path, "wb") as f:
        f.write(struct.pack("<6I", MAGIC, world, 0, 0, attached, 0) + b"\0" * 40)
        f.write(struct.pack("<32Q", *([0x1234567] * 32)))        # left-over hello / ack words
        f.truncate(SEG_BYTES)
    return path


@pytest.mark.parametrize("attached", [0, 1, 2])
def test_stale_segment_is_never_joined(L, attached):
    name = f"/mcg_pytest_stale_{os.getpid()}_{attached}"
    path = _plant_stale_segment(name, 2, attached)
    stale_ino = os.stat(path).st_ino
    res = {}
    t1 = threading.Thread(target=_attach, args=(L, name, 2, 1, 30.0, res, 1))
    t1.start()                       # rank 1 is quicker than rank 0: it finds and maps the stale segment ...
    time.sleep(0.5)
    assert t1.is_alive()             # ... and must not consider the job complete on its own
    t0 = threading.Thread(target=_attach, args=(L, name, 2, 0, 30.0, res, 0))
    t0.start()                       # rank 0 replaces it with the job's own
    t0.join(40)
    t1.join(40)
    assert not t0.is_alive() and not t1.is_alive()
    assert res[0][0] == 0 and res[1][0] == 0, res
    assert os.stat(path).st_ino != stale_ino
    # both ranks sit on the SAME fresh segment: a barrier between them completes
    out = {}
    ts = [threading.Thread(target=lambda r=r: out.__setitem__(r, L.mcg_debug_shm_barrier(res[r][1]))) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert out == {0: 0, 1: 0}
    L.mcg_debug_shm_detach(res[1][1])
    L.mcg_debug_shm_detach(res[0][1])
    assert not os.path.exists(path)


def test_wrong_rank_count_is_an_error(L):
    name = f"/mcg_pytest_count_{os.getpid()}"
    res = {}
    t0 = threading.Thread(target=_attach, args=(L, name, 3, 0, 3.0, res, 0))   # a job of three ...
    t0.start()
    time.sleep(0.3)
    _attach(L, name, 2, 1, 3.0, res, 1)                                        # ... joined by a rank that expects two
    t0.join(10)
    assert res[1][0] == 7 and "created for 3 ranks" in res[1][2]
    assert res[0][0] == 7 and "attached" in res[0][2]                          # rank 0 gives up after its time-out
    assert not os.path.exists("/dev/shm" + name)


def test_barrier_time_out_poisons_the_segment(L):
    """One rank never arrives: the waiting rank gives up after its time-out AND marks the segment, so the late rank (and
    everybody else, from then on) fails at once instead of running out of step."""
    name = f"/mcg_pytest_poison_{os.getpid()}"
    res = {}
    ts = [threading.Thread(target=_attach, args=(L, name, 2, r, 1.0, res, r)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    assert res[0][0] == 0 and res[1][0] == 0
    t0 = time.time()
    assert L.mcg_debug_shm_barrier(res[0][1]) == 7            # MCG_ERR_COMM after ~1 s
    assert 0.5 < time.time() - t0 < 10
    t0 = time.time()
    assert L.mcg_debug_shm_barrier(res[1][1]) == 7            # at once
    assert time.time() - t0 < 0.5
    assert b"poisoned" in L.mcg_last_error()
    L.mcg_debug_shm_detach(res[1][1])
    L.mcg_debug_shm_detach(res[0][1])


def test_poison_releases_a_waiting_rank(L):
    """A rank that fails locally between two collective steps poisons the segment: a peer already waiting in the
    barrier returns with an error within milliseconds."""
    name = f"/mcg_pytest_poison2_{os.getpid()}"
    res = {}
    ts = [threading.Thread(target=_attach, args=(L, name, 2, r, 60.0, res, r)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(20)
    out = {}
    w = threading.Thread(target=lambda: out.__setitem__("rc", L.mcg_debug_shm_barrier(res[0][1])))
    w.start()
    time.sleep(0.3)
    assert w.is_alive()
    assert L.mcg_debug_shm_poison(res[1][1]) == 0
    w.join(5)
    assert not w.is_alive() and out["rc"] == 7
    L.mcg_debug_shm_detach(res[1][1])
    L.mcg_debug_shm_detach(res[0][1])


@pytest.mark.parametrize("world", [8, 16])
def test_eight_and_sixteen_ranks_join_and_stay_in_step(L, world):
    """BASELINE.json configs[4]'s world size (and the segment's maximum): the ranks arrive in a scrambled order, rank 0 in
    the middle; everybody joins the same fresh segment; 300 barriers keep them in step (no rank is ever seen a round
    ahead of another between two barriers); a poison while fifteen of sixteen wait releases all of them at once."""
    name = f"/mcg_pytest_w{world}_{os.getpid()}"
    res, order = {}, [(5 * r + 3) % world for r in range(world)]
    assert sorted(order) == list(range(world))
    ts = []
    for r in order:
        t = threading.Thread(target=_attach, args=(L, name, world, r, 60.0, res, r))
        t.start()
        ts.append(t)
        time.sleep(0.01)
    for t in ts:
        t.join(90)
    assert all(res[r][0] == 0 for r in range(world)), res
    rounds, at, errors = 300, [0] * world, []

    def run(r):
        for k in range(rounds):
            at[r] = k
            if L.mcg_debug_shm_barrier(res[r][1]) != 0:
                errors.append((r, k, "barrier failed"))
                return
            lo, hi = min(at), max(at)          # everybody has reached round k; nobody can be beyond k + 1
            if lo < k or hi > k + 1:
                errors.append((r, k, lo, hi))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts) and not errors, errors[:5]
    out = {}
    ws = [threading.Thread(target=lambda r=r: out.__setitem__(r, L.mcg_debug_shm_barrier(res[r][1]))) for r in range(1, world)]
    for w in ws:
        w.start()
    time.sleep(0.3)
    assert all(w.is_alive() for w in ws)                 # world - 1 ranks wait for the last one ...
    assert L.mcg_debug_shm_poison(res[0][1]) == 0        # ... which fails locally instead
    t0 = time.time()
    for w in ws:
        w.join(10)
    assert time.time() - t0 < 5 and out == {r: 7 for r in range(1, world)}
    for r in reversed(range(world)):
        L.mcg_debug_shm_detach(res[r][1])
    assert not os.path.exists("/dev/shm" + name)
