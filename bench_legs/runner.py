"""Launching and fault isolation of bench.py: the N-rank launcher (no GPU call in the parent), LegRunner (budgeted, fault-isolated
secondary legs), the one JSON line, the exit codes, and the CPU self-test of all of it (tests/test_host_cpu.py)."""
import json
import os
import socket
import subprocess
import sys
import time


BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(n, argv):
    """Parent of an N-rank run: nothing here may touch the GPU (a process that has initialised HIP must not exec or be
    replaced, and the children need the devices).  Children inherit stderr; rank 0's JSON line is the last stdout line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH_PY] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    return res.returncode if res.returncode else (0 if js else 1)


class LegRunner:
    """Secondary legs of the JSON line, fault-isolated: whatever happens in one of them -- an exception on this rank, an exception
    on ANOTHER rank that leaves this one inside a collective, a collective that never completes -- the headline fields the timed
    loop has already earned are printed and the leg carries {"error": ...}.  Exit code: 0 while the ranks stayed in step (a leg that
    raised on EVERY rank alike, a non-collective leg that raised); EXIT_OUT_OF_STEP (3) when a watchdog ended a hung leg or the job
    could not be torn down in order -- a hung collective or a GPU fault must not look like success to torchrun / the driver.  The
    line is on stdout before any rank leaves non-zero (`leave_out_of_step`).

    * An exception is caught, recorded in the leg and announced to the other ranks through the process group's key-value store (no
      collective: the ranks are no longer in step); collective legs that have not started yet are skipped everywhere.
    * A wall-clock budget per leg, kept by a watchdog thread on every rank: when it runs out, rank 0 prints the line as it stands
      (the running leg marked as timed out) and every rank leaves with os._exit(EXIT_OUT_OF_STEP) -- a process that has touched the
      GPU exits, it never re-executes anything.  (torch's own NCCL watchdog would abort the whole process group with SIGABRT instead: its
      timeout is set beyond the budgets here, init_process_group(timeout=...).)"""

    def __init__(self, out, rank, store=None, emit=None):
        import threading
        self.out, self.rank, self.store, self.emit = out, rank, store, emit
        self.lock = threading.Lock()
        self.current = None          # (name, deadline, budget)
        self.failed_here = False
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def _watch(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                cur = self.current
                if cur is None or time.monotonic() < cur[1]:
                    continue
                if self.rank == 0 and self.out is not None:
                    self.out[cur[0]] = {"error": f"leg exceeded its wall-clock budget of {cur[2]:.0f} s (a rank failed or a collective hung); "
                                                 "the legs after it were not run"}
                    self.emit(self.out)
            self._announce()
            leave_out_of_step(self.store, self.rank)

    def _announce(self):
        try:
            if self.store is not None:
                self.store.add("cmdiad_bench_leg_failed", 1)
        except Exception:
            pass

    def others_failed(self):
        try:
            return self.store is not None and self.store.add("cmdiad_bench_leg_failed", 0) > 0
        except Exception:
            return True

    @property
    def in_step(self):
        """False once any rank has failed a leg: the ranks may be at different points, no further collective is safe."""
        return not self.failed_here and not self.others_failed()

    def run(self, name, fn, budget_s, collective=False):
        if collective and not self.in_step:
            res = {"skipped": "an earlier leg failed on some rank: the ranks are no longer in step, collective legs are skipped"}
        else:
            with self.lock:
                self.current = (name, time.monotonic() + budget_s, budget_s)
            try:
                res = fn()
            except Exception as e:          # noqa: BLE001 -- a secondary leg must never cost the headline
                import traceback
                traceback.print_exc(file=sys.stderr)
                res = {"error": f"{type(e).__name__}: {e}"[:600]}
                self.failed_here = True
                self._announce()
            finally:
                with self.lock:
                    self.current = None
        if self.out is not None and res is not None:
            with self.lock:
                self.out[name] = res
        return res


EXIT_OUT_OF_STEP = 3     # the line was printed, but a leg hung / a rank failed and the ranks could not be torn down in order
_LINE_KEY = "cmdiad_bench_line_printed"


def leave_out_of_step(store, rank, wait_s=None):
    """End this rank without an orderly teardown (the other ranks may sit in a collective that never completes) with a NON-ZERO
    exit code.  torch.distributed.run terminates the surviving ranks as soon as one exits non-zero, so a rank other than 0 first
    waits (bounded) until rank 0 has put the JSON line on stdout; rank 0 raises that flag here, after its emit_line."""
    import datetime
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        if store is not None:
            if rank == 0:
                store.set(_LINE_KEY, "1")
            else:
                if wait_s is None:
                    wait_s = float(os.environ.get("CMDIAD_BENCH_LEAVE_WAIT", "0") or 0) or 660.0   # beyond the longest leg budget
                store.wait([_LINE_KEY], datetime.timedelta(seconds=wait_s))
    except Exception:       # the store went away with rank 0: nothing left to wait for
        pass
    os._exit(EXIT_OUT_OF_STEP)


def emit_line(out):
    """THE one JSON line.  RCCL writes its version banner to C stdout, which is flushed at exit -- i.e. AFTER a Python print: push it
    out first so that the JSON line is the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def selftest_launch():
    """CPU self-test of the launch path (tests/test_host_cpu.py): every rank joins a gloo group and runs the row-sharded merge
    (engine.gather_queries + engine.merge_shard_keys) on host tensors; no GPU call anywhere.  The merge and two more collective
    steps run as LegRunner legs, with CMDIAD_BENCH_INJECT="<leg>:<rank>:<raise|hang>" injecting a failure: rank 0 must still print
    one JSON line with the headline fields intact."""
    import datetime
    import torch
    import torch.distributed as td
    from cmdiad_amd import engine as eng
    td.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    rank, world = td.get_rank(), td.get_world_size()
    inject = (os.environ.get("CMDIAD_BENCH_INJECT") or "::").split(":")
    budget = float(os.environ.get("CMDIAD_BENCH_LEG_BUDGET", "20"))
    out = {"selftest_launch": True, "ranks": world, "metric": "selftest", "value": 1.0} if rank == 0 else None
    legs = LegRunner(out, rank, td.distributed_c10d._get_default_store(), emit_line)

    def maybe_fail(name):
        if inject[0] == name and int(inject[1]) == rank:
            if inject[2] == "raise":
                raise RuntimeError(f"injected failure in {name} on rank {rank}")
            time.sleep(3600)

    def merge():
        maybe_fail("merge")
        g = torch.Generator().manual_seed(5)
        Q, Nb = 64, 1000
        d2 = torch.rand(Q, Nb, generator=g)                        # the same on every rank
        lo, hi = eng.shard_range(Nb, rank, world)
        keys = torch.full((Q,), eng.KEY_EMPTY, dtype=torch.int64)
        if hi > lo:
            v, i = d2[:, lo:hi].min(1)
            keys = (v.view(torch.int32).to(torch.int64) << 32) | (i + lo)
        q16 = torch.full((4, 8), float(rank), dtype=torch.float16)
        q_all, s_all = eng.gather_queries(q16, torch.full((4,), float(rank)), td.group.WORLD)
        keys = eng.merge_shard_keys(keys, td.group.WORLD)
        ok = bool(torch.equal(keys & 0xFFFFFFFF, d2.argmin(1))) and q_all.shape[0] == 4 * world \
            and bool(torch.equal(s_all, torch.arange(world, dtype=torch.float32).repeat_interleave(4)))
        flag = torch.tensor([1 if ok else 0])
        td.all_reduce(flag, op=td.ReduceOp.MIN)
        return {"merge_ok": bool(flag.item())}

    def count(name):
        def fn():
            maybe_fail(name)
            t = torch.ones(1)
            td.all_reduce(t)
            return {"ranks_counted": int(t.item())}
        return fn

    res = legs.run("merge", merge, budget, collective=True)
    legs.run("second", count("second"), budget, collective=True)
    legs.run("third", count("third"), budget, collective=True)
    if legs.in_step:
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        out["merge_ok"] = bool(res.get("merge_ok", False))
        emit_line(out)
        sys.stdout.flush()
    if not legs.in_step:
        leave_out_of_step(legs.store, rank)   # the other ranks may sit in a collective that will never complete: no orderly teardown
    return 0 if res.get("merge_ok") else 1


# --------------------------------------------------------------------------------------------------------- state
