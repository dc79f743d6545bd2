"""bench.py's N > 1 machinery, moved here UNCHANGED in round 6 (no new modes, guardians or rehearsals
until a node run exists; bench.py's timed path should read on one screen): the guardian process that owns the job's one JSON
line (LastWill), the collective set-up by agreement (install_collective, rccl_forms_in_time), and the drivers of the C5 rows
of an N > 1 run (c5_sharded_rows, in a child job or inline).  What each does, and why, is said in its docstring; DESIGN.md
section 6 has the protocol.  tests/test_bench_rehearsal.py runs all of it at eight CPU ranks through `bench.py --rehearsal`."""
from __future__ import annotations

import json
import os
import sys
import time

from tools.bench_common import DT, RB, ROOT, SEED


class Gpu:
    """What this script asks of torch.cuda (tests/bench_rehearsal.py has the CPU stand-in of --rehearsal)."""
    name = "cuda"

    @staticmethod
    def set_device(d):
        import torch
        torch.cuda.set_device(d)

    @staticmethod
    def synchronize():
        import torch
        torch.cuda.synchronize()

    @staticmethod
    def current_stream_handle():
        import torch
        return torch.cuda.current_stream().cuda_stream

    @staticmethod
    def device_count():
        import torch
        return torch.cuda.device_count()


def agree(flags, dist, torch, dev) -> list:
    """Element-wise AND of `flags` over the ranks (one all-reduce).  Every rank enters it -- from its except branch too."""
    t = torch.tensor([1 if f else 0 for f in flags], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [int(x) == 1 for x in t.tolist()]


def everyone(ok: bool, dist, torch, dev) -> bool:
    """True iff `ok` on EVERY rank.  Every rank enters it -- from its except branch too -- so it doubles as the point where
    the ranks of a step that may fail locally meet again.  That is sound only where the step itself cannot leave a peer
    inside ANOTHER collective for good: set-up steps (nothing collective inside), and passes over the node mailbox (shm /
    ipc: the segment barrier times out, the peers raise and arrive here too).  A pass over the built-in RCCL communicator or
    over torch.distributed is not bounded like that -- see c5_sharded_rows.phase for what a failing rank does there."""
    return agree([ok], dist, torch, dev)[0]


class LastWill:
    """Rank 0 of an N > 1 run: a guardian process, forked before anything touches the GPU, that owns the job's ONE JSON
    line.  The rank sends it the line as soon as the headline is complete ("WILL", re-sent after every C5 row) and the
    finished line at the end ("FINAL"); when the pipe closes -- the rank returned, raised, was killed by the launcher after
    a peer died, or took a device fault in a C5 row -- the guardian prints FINAL, or else the last WILL with an "aborted"
    note.  Nothing after the headline can cost the line any more, whatever the C5 rows do."""

    def __init__(self, json_out):
        import signal
        r, w = os.pipe()
        self.pid = os.fork()
        if self.pid == 0:
            code = 0
            try:
                os.close(w)
                for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                    signal.signal(sg, signal.SIG_IGN)
                will = final = None
                with os.fdopen(r, "r") as f:
                    for ln in f:
                        if not ln.endswith("\n"):
                            break                      # (the rank died inside a write)
                        if ln.startswith("WILL "):
                            will = ln[5:]
                        elif ln.startswith("FINAL "):
                            final = ln[6:]
                line = final
                if line is None and will is not None:
                    j = json.loads(will)
                    j["aborted"] = ("rank 0 ended before the line was finished (a peer rank failed and the launcher ended the job, or "
                                    "a C5 row took the process down); the headline was complete, extra.configs holds the rows that were")
                    line = json.dumps(j)
                if line:
                    json_out.write(line.rstrip("\n") + "\n")
                    json_out.flush()
            except BaseException:   # noqa: BLE001
                code = 1
            finally:
                os._exit(code)
        os.close(r)
        self._w = os.fdopen(w, "w")
        self._out, self._lost = json_out, False

    def _send(self, tag: str, out: dict) -> None:
        try:
            self._w.write(tag + " " + json.dumps(out) + "\n")
            self._w.flush()
        except OSError:   # the guardian is gone (it should never be): this process prints the line itself at the end
            self._lost = True

    def update(self, out: dict):
        if not self._lost:
            self._send("WILL", out)

    def final(self, out: dict):
        if not self._lost:
            self._send("FINAL", out)
        try:
            self._w.close()
        except OSError:
            pass
        _, status = os.waitpid(self.pid, 0)
        if self._lost or status != 0:
            print(json.dumps(out), file=self._out, flush=True)


# Set when a scratch context was left inside ncclCommInitRank by rccl_forms_in_time: the daemon thread still holds a GPU context
# and a half-formed communicator, and library destructors at interpreter exit could block on it -- bench.py then says so in its
# line (config.rccl_init_abandoned) and leaves through os._exit once the line is out.
ABANDONED = {"rccl_init": False}

GPU_PROCESS_GUARD = 6   # what the pool's process guard allowed on the builder's one-GPU box; a node's is not stated


def c5_rows_mode(args, world: int) -> str:
    """child: every rank starts a child process for the C5 rows while it still holds the device (2 x world GPU processes; a
    fault in a row cannot touch the parent).  inline: the rows run in the rank processes, after the headline is safe with
    the guardian (world GPU processes).  auto: child where 2 x world fits under the process guard measured, else inline."""
    if args.c5_rows != "auto":
        return args.c5_rows
    return "child" if 2 * world <= GPU_PROCESS_GUARD else "inline"


def rccl_forms_in_time(make_scratch, rank: int, world: int, bcast, limit_s: float) -> bool:
    """Form the built-in RCCL communicator on a scratch context inside a time box; True iff it formed within limit_s.  Only
    ncclCommInitRank itself runs in the worker thread: the id's broadcast stays in the MAIN thread (torch's current device is
    thread-local -- a collective issued from a fresh thread would run on device 0 on every rank), which enters it ALWAYS,
    whatever the worker did (an empty id from rank 0 makes every rank's init raise together)."""
    import threading
    box = {}
    uid_ready, uid_back = threading.Event(), threading.Event()

    def bcast_in_main(uid):   # called by init_rccl inside the worker: park the id, wait for the main thread's broadcast
        box["uid_in"] = uid
        uid_ready.set()
        uid_back.wait()
        return box.get("uid_out")

    def work():
        try:
            e = make_scratch()
            box["scratch"] = e
            e.init_rccl(rank, world, bcast_in_main)
            box["formed"] = True
        except Exception as ex:   # noqa: BLE001
            box["err"] = ex
        finally:
            uid_ready.set()       # (a worker that failed before it had an id must not keep the main thread from the broadcast)
    t = threading.Thread(target=work, daemon=True)
    t.start()
    uid_ready.wait(limit_s)       # creating the id is local
    box["uid_out"] = bcast(box.get("uid_in", b"") if rank == 0 else None)
    uid_back.set()
    t.join(limit_s)
    if t.is_alive():
        ABANDONED["rccl_init"] = True
        print(f"bench: rank {rank}: the built-in RCCL communicator did not form within {limit_s:.0f} s; falling back to torch.distributed",
              file=sys.stderr, flush=True)
        return False
    if "err" in box or not box.get("formed"):
        print(f"bench: built-in RCCL communicator unavailable ({box.get('err')}); using torch.distributed", file=sys.stderr)
        if box.get("scratch") is not None:
            try:
                box["scratch"].close()
            except Exception:   # noqa: BLE001
                pass
        return False
    box["scratch"].close()
    return True


def install_collective(eng, mc, want: str, dist, torch, rank: int, world: int, dev="cuda", scratch=None) -> str:
    """Give `eng` the collective `want` ("ipc", "shm", "rccl", "torch") -- every rank ends up on the SAME one: a set-up
    that fails on any rank sends all of them one step down (ipc -> shm -> rccl -> torch).  Returns what is installed.
    No rank can be left alone in a collective: whatever a rank does before a broadcast cannot fail (the segment's name is
    a string; the RCCL id is created inside a try and an empty one is broadcast on failure, PathEngine.init_rccl), and
    every local step that can fail is followed by everyone()."""
    got = want
    if want in ("shm", "ipc"):
        box = [f"/mcg_bench_{os.getpid()}_{time.time_ns()}" if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ok = True
        try:
            eng.init_shm(box[0], rank, world)
        except mc.McgError as e:
            print(f"bench: shared-memory communicator unavailable ({e}); using RCCL", file=sys.stderr)
            ok = False
        if not everyone(ok, dist, torch, dev):
            eng.set_allreduce(None)
            got = f"rccl ({want} init failed" + ("" if not ok else " on a peer") + ")"
        elif want == "ipc":
            try:     # (collective over the segment: the ranks agree inside; an error poisons the segment for all of them)
                peer = eng.shm_peer_mailbox(True)
            except mc.McgError as e:
                print(f"bench: peer-memory mailbox failed ({e})", file=sys.stderr)
                peer, ok = False, False
            if not everyone(ok, dist, torch, dev):
                eng.set_allreduce(None)
                got = "rccl (ipc set-up failed" + ("" if not ok else " on a peer") + ")"
            elif not peer:
                got = "shm (peer-memory mailbox unavailable: export, open or in-kernel ping failed on some rank)"
    if got.startswith("rccl"):
        def bcast(uid):
            box = [uid]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        ok = True
        try:                               # can EVERY rank load librccl?  (agreed before anybody enters ncclCommInitRank, where a
            eng.rccl_probe()               #  rank whose peer never arrives would wait)
        except mc.McgError as e:
            print(f"bench: librccl unavailable on rank {rank} ({e})", file=sys.stderr)
            ok = False
        if everyone(ok, dist, torch, dev):
            # ncclCommInitRank with more than one rank has never run in this repo's history (every GPU box had one GPU): a
            # communicator that does not FORM must cost the run its collective, not its line.  So it is formed once on a
            # scratch context inside a time box; only if every rank's formed in time does the real context get its own.  A
            # scratch context that is still inside ncclCommInitRank when the box closes is abandoned (daemon thread).
            ok = rccl_forms_in_time(scratch or (lambda: mc.PathEngine(eng.device)), rank, world, bcast,
                                    float(os.environ.get("MCG_BENCH_RCCL_INIT_LIMIT", "90")))
            if everyone(ok, dist, torch, dev):
                try:
                    eng.init_rccl(rank, world, bcast)
                except mc.McgError as e:       # communicator set-up failed on this node: use torch's, and say so
                    print(f"bench: built-in RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
                    ok = False
            else:
                ok = False
        else:
            ok = False
        if not everyone(ok, dist, torch, dev):               # all ranks take the same route
            got = "torch (built-in RCCL init failed" + ("" if not ok else " on a peer") + ")"
            eng.use_torch_distributed()
    elif got == "torch":
        eng.use_torch_distributed()
    return got


def c5_sharded_rows(args, mc, N, dist, torch, device, stream, rank, world, hw=Gpu, make_engine=None, on_row=None,
                    deadline=None) -> list:
    """BASELINE.json configs[4] on this run's N ranks, after the headline loop: rBergomi (H = 0.1, eta = 1.9) American put,
    LSM order 2, 252 steps, --c5-paths (8M) paths per GPU of ONE Philox stream, timed through each collective of
    --c5-collectives in turn on a fresh context.  One untimed pass, then 3 timed between barriers; per row: the slowest
    and the fastest rank's ms per pass, the collective that ran, what its communicator has seen (mcg_comm_info), the
    launches of the LSM sweep per pass (1 = the one-launch sweep exchanged inside the kernel) and the global price.
    A row is a sequence of local phases; after each the ranks meet in agree(): a rank that raised is there too, so the
    row is recorded as failed on ALL ranks at once -- WHERE the peers can get there: set-up phases and passes over the node
    mailbox (its barrier times out).  A rank that raises inside a pass over the built-in RCCL communicator or over
    torch.distributed leaves its peers inside an all-reduce that never completes (or, worse, would pair its own agreement
    all-reduce with their data all-reduce): there it ends the job instead -- exit code 17, the launcher (or the parents of
    the child job) take the peers down, the rows finished so far and the headline are already with rank 0's guardian.
    `deadline` (time.time() value): once any rank is past it the remaining rows are abandoned by all ranks together."""
    from montecarlooptionspricer_amd.sharding import shard_range
    rows, reps, steps = [], 3, 252
    total = args.c5_paths * world
    begin, count = shard_range(total, rank, world, align=2)
    dev = torch.device("cuda", device) if hw is Gpu else torch.device("cpu")
    make_engine = make_engine or (lambda: mc.PathEngine(device, stream=stream))
    out_of_time = False
    for want in [c for c in args.c5_collectives.split(",") if c]:
        if out_of_time:
            break
        e5, err, row = None, None, None
        timeouts_before = mc.stats()["lsm_one_launch_timeouts"]   # (process-wide counter: the row reports its own share)
        if args.rehearsal:
            os.environ["MCG_REHEARSAL_ROW"] = want   # (read by the rehearsal's failure injection only)
        st = {"unbounded": False}

        def phase(fn):
            """Run a local step; every rank then learns whether it worked everywhere (and whether there is time left)."""
            nonlocal err, out_of_time
            ok = True
            if err is None:
                try:
                    fn()
                except Exception as ex:   # noqa: BLE001
                    err, ok = f"{type(ex).__name__}: {ex}", False
                    if st["unbounded"]:
                        print(f"bench: rank {rank} failed inside a pass over '{st.get('got')}' ({err}); its peers cannot leave that "
                              "collective, so this rank ends the job (exit code 17)", file=sys.stderr, flush=True)
                        os._exit(17)
            else:
                ok = False
            in_time = deadline is None or time.time() < deadline
            ok, in_time = agree([ok, in_time], dist, torch, dev)
            if not in_time:
                out_of_time = True
            return ok and in_time

        def setup():
            nonlocal e5
            e5 = make_engine()

        def one_pass():
            P = e5.rbergomi(SEED, RB["S0"], RB["r"], RB["xi"], RB["H"], RB["eta"], RB["rho"], DT, steps, count, path_begin=begin)
            r = e5.price_lsm(P, RB["r"], 100.0, steps * DT, DT, False, 2)
            P.free()
            return r

        def warm():
            one_pass()
            e5.synchronize()
            hw.synchronize()

        def timed():
            e5.timing_enable(True)
            e5.timing_reset()
            t0 = time.perf_counter()
            for _ in range(reps):
                st["price"], st["se"] = one_pass()
            e5.synchronize()
            hw.synchronize()
            st["mine"] = (time.perf_counter() - t0) / reps * 1e3

        try:
            good = phase(setup)
            if good:
                # (install_collective agrees among the ranks inside; an exception there is the same on every rank)
                st["got"] = "none (every rank prices its own shard alone: a local price, the baseline the routes below add their exchange to)" \
                    if want == "none" else install_collective(e5, mc, want, dist, torch, rank, world, dev, scratch=make_engine)
                st["info"] = e5.comm_info()
                st["unbounded"] = st["got"].startswith(("rccl", "torch"))
            good = good and phase(warm) and phase(timed)
            st["unbounded"] = False
            if good:
                t = torch.tensor([st["mine"], -st["mine"]], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms_max, ms_min = float(t[0].item()), -float(t[1].item())
                gen_ms, _ = e5.timing_get(N.K_RBERGOMI)
                sw_ms, sw_n = e5.timing_get(N.K_LSM_SWEEP)
                seen = torch.tensor([st["info"]["seen_ranks"]], device=dev)
                dist.all_reduce(seen, op=dist.ReduceOp.MIN)
                row = {
                    "config": f"C5: rBergomi American put LSM order 2, {args.c5_paths} paths x {steps} steps per GPU, {world} rank(s) "
                              f"= {total} paths of one Philox stream",
                    "collective_requested": want, "collective": st["got"],
                    "comm": dict(st["info"], seen_ranks_min_over_ranks=int(seen.item())),
                    "paths_per_gpu": args.c5_paths, "global_paths": total,
                    "ms_per_pass_slowest_rank": ms_max, "ms_per_pass_fastest_rank": ms_min,
                    "Mpaths_per_s": total / ms_max / 1e3, "price": st["price"], "std_err": st["se"],
                    "rank0_generator_ms_per_pass": gen_ms / reps, "rank0_lsm_sweep_ms_per_pass": sw_ms / reps,
                    "rank0_lsm_sweep_launches_per_pass": sw_n // reps,
                    "lsm_one_launch": e5.lsm_one_launch_enabled() and sw_n // reps <= 2,
                    # one-launch sweeps of THIS row whose hand-shake gave up on rank 0 (two rank processes on one card cannot both
                    # be resident): > 0 means the row timed the per-date fall-back, not the mailbox sweep
                    "lsm_one_launch_timeouts": mc.stats()["lsm_one_launch_timeouts"] - timeouts_before,
                    "rank0_stats": mc.stats()}
            elif out_of_time:
                row = {"config": "C5", "collective_requested": want,
                       "error": "the wall-clock budget of the C5 rows was used up: this row and the remaining ones were abandoned by all ranks together"}
            else:
                row = {"config": "C5", "collective_requested": want,
                       "error": err or "a peer rank failed in this row (its own stderr says why); all ranks abandoned it together"}
        except Exception as ex:   # (outside the phases: the collectives of this function itself)
            row = {"config": "C5", "collective_requested": want, "error": f"{type(ex).__name__}: {ex}"}
        finally:
            if e5 is not None:
                e5.close()
        rows.append(row)
        if on_row is not None:
            on_row(rows)
    return rows


def c5_rows_in_child_job(args, dist, torch, rank: int, world: int, budget_s: float):
    """Every rank of this job starts `bench.py --c5-child` as a child process (same RANK / LOCAL_RANK / WORLD_SIZE, a
    rendezvous port of its own) and the parents WATCH the children together: four times a second they exchange (over a gloo
    group of their own: no device work beside the children's timing) who is still running and who has failed -- a child
    that could not be started (spawn refused), one that exited non-zero, or the budget running out.  On the first failure
    every parent kills its child: no parent waits for a child whose peer is gone.  Rank 0's child prints each finished row
    as a line of its own, so the rows before a failure are kept.  Returns (on rank 0) the rows plus, after a failure, one
    row that says what went wrong."""
    import socket
    import subprocess
    import tempfile
    from datetime import timedelta
    mon = dist.new_group(backend="gloo", timeout=timedelta(seconds=120))
    box = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            box[0] = sk.getsockname()[1]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(box[0]),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in list(env):   # the child is not an elastic worker of the parent's agent
        if k.startswith("TORCHELASTIC_") or k in ("TORCH_NCCL_ASYNC_ERROR_HANDLING",):
            env.pop(k)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--c5-child", "--gpus", str(world), "--backend", args.backend,
           "--c5-paths", str(args.c5_paths), "--c5-collectives", args.c5_collectives, "--c5-budget", str(budget_s)] + (["--rehearsal"] if args.rehearsal else [])
    fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
    proc, why = None, None
    try:
        if os.environ.get("MCG_BENCH_SPAWN_FAIL") in (str(rank), "all"):   # test hook: the pool refuses the process
            raise OSError(11, "Resource temporarily unavailable (injected)")
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, text=True)
    except OSError as e:
        why = f"rank {rank}: the child process could not be started ({e})"
        print("bench: " + why, file=sys.stderr, flush=True)
    t_end = time.time() + budget_s + 60.0   # (the child abandons its rows at budget_s by itself; this is for one that hangs)
    failed_any = False
    while True:
        rc = proc.poll() if proc is not None else 1
        failed = proc is None or (rc is not None and rc != 0) or time.time() > t_end
        t = torch.tensor([1 if failed else 0, 1 if (proc is not None and rc is None) else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=mon)
        if int(t[0]):
            failed_any = True
            if proc is not None and proc.poll() is None:
                proc.kill()
            break
        if not int(t[1]):
            break
        time.sleep(0.25)
    if proc is not None:
        try:
            proc.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pass
    rows = None
    if rank == 0:
        fo.seek(0)
        rows = [json.loads(ln[4:]) for ln in fo.read().splitlines() if ln.startswith("ROW ")]
        if failed_any:
            fe.seek(0)
            rc = proc.returncode if proc is not None else None
            rows.append({"config": "C5", "error": why or (f"child job failed (exit code {rc})" if rc not in (None, 0, -9) else
                                                          "child job ended by its parents: a peer rank's child failed, could not be started, or "
                                                          f"the job ran past {budget_s + 60:.0f} s (every rank's own stderr says which)"),
                         "stderr_tail": fe.read()[-1500:]})
    dist.barrier()
    return rows


def c5_rows_inline(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, will, out, budget_s: float):
    """The C5 rows in the rank processes themselves (world GPU processes, not 2 x world).  The headline is with the guardian
    already; every finished row is sent after it.  A wall-clock budget: past it the ranks abandon the remaining rows
    together (checked in every agreement); a rank still inside a row a minute after that -- a collective that never
    returns -- ends the job (exit code 18), which the guardian's line survives."""
    import threading
    dog = threading.Timer(budget_s + 60.0, lambda: (print(f"bench: rank {rank}: the inline C5 rows hang past their budget; ending the job",
                                                          file=sys.stderr, flush=True), os._exit(18)))
    dog.daemon = True
    dog.start()

    def on_row(rows):
        if will is not None:
            out.setdefault("extra", {})["configs"] = list(rows)
            will.update(out)
    try:
        return c5_sharded_rows(args, mc, N, dist, torch, device, stream, rank, world, hw, make_engine, on_row=on_row,
                               deadline=time.time() + budget_s)
    finally:
        dog.cancel()
