"""GPU: the row-sharded paths with TWO REAL RANKS.  Every other multi-rank test of this suite is either a "fake world" (W shards
searched in turn by one process), an RCCL world of ONE rank, or a gloo world on the CPU with a torch stand-in for the kernels: none
runs the HIP kernels inside a process group of more than one rank.  Here two processes share the box's one GPU (RCCL refuses two
ranks on one device, so the collectives travel over gloo; the kernels, the shard arithmetic, the per-rank queries, the counts
exchange, the MIN / MAX / SUM reductions and the pipeline's repeat-on-overflow are the real ones): tests/world2_worker.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_sharded_paths_with_two_real_ranks_on_one_gpu():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("OMP_NUM_THREADS", "8")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "tests", "world2_worker.py")],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["world2"] is True and rec["ranks"] == 2
    assert [c.split("(")[0] for c in rec["checked"]] == ["sharded_search", "sharded_fp32_scoring", "sharded_coreset", "sharded_pipeline",
                                                          "sharded_fp32_pipeline"]


_CLASS_METRICS = {}     # (world, bank) -> per-class metric tuples of the bench's class loop, compared across the parametrised runs


def _launch(nproc, script_args, extra_env, timeout=900):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(extra_env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), *script_args], capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)


@pytest.mark.slow
@pytest.mark.parametrize("world,bank", [pytest.param(2, "replicated", marks=pytest.mark.rehearsal), (2, "sharded"),
                                        pytest.param(8, "sharded", marks=pytest.mark.rehearsal)])
def test_bench_line_of_n_ranks_rehearsed_on_one_gpu(world, bank):
    """The driver's N > 1 command (`torch.distributed.run --nproc-per-node N bench.py --gpus N`) with every rank on device 0 and gloo
    in RCCL's place (CMDIAD_BENCH_ONE_DEVICE=1): the census, the max-over-ranks timing, every collective leg (sharded_search with its
    three collectives per search, the class loop dealt to N ranks, the teardown) run with a world of N REAL ranks -- N = 8 is the
    driver's largest -- and rank 0 prints the one line with no leg in error."""
    import time
    t0 = time.perf_counter()
    out = _launch(world, [os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "2", "--bank", bank,
                          "--classes", "cookie,bagel", "--class-scale", "0.02", "--class-test", "10"],
                  {"CMDIAD_BENCH_ONE_DEVICE": "1", "OMP_NUM_THREADS": "2"}, timeout=1500)
    print(f"[bench.py --gpus {world} --bank {bank}, {world} ranks on one device: {time.perf_counter() - t0:.0f} s]")
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["world"] == world and d["rccl_ranks"] == world and d["backend"] == "gloo" and "rehearsal" in d
    assert [r["rank"] for r in d["ranks"]] == list(range(world)) and len({r["pid"] for r in d["ranks"]}) == world
    assert d["scaling"] == "weak" and abs(d["value"] - world * 32 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    assert not [k for k, v in d.items() if isinstance(v, dict) and ("error" in v or "skipped" in v)], {k: v for k, v in d.items() if isinstance(v, dict) and "error" in v}
    s = d["sharded_search"]
    assert s["rccl_ranks"] == world and [c["cls"] for c in s["classes"]] == ["cookie", "bagel"]
    for c in s["classes"]:
        assert len(c["live_rows_per_rank"]) == world and c["rows_this_rank"] < c["rows"] and c["gather_MB_received_per_rank"] > 0
    m = d["mtfi_classes"]
    assert m["world"] == world and len(m["per_class"]) == 10 and len(m["assignment"]) == world
    # a class is evaluated start to finish on one rank: its metrics cannot depend on how many ranks there are, which rank got it,
    # or whether its host SVM fits ran beside another class's device work (they do when a rank has more than one class)
    metrics = {c: tuple(v[k] for k in ("image_rocauc", "pixel_rocauc", "au_pro", "au_pro_001") if k in v) for c, v in m["per_class"].items()}
    assert all(len(t) >= 3 for t in metrics.values())
    for other_world, other in _CLASS_METRICS.items():
        assert other == metrics, f"per-class metrics differ between {other_world} and {world} ranks"
    _CLASS_METRICS[(world, bank)] = metrics
    assert ("row-sharded" in d["config"]["bank"]) == (bank == "sharded")
