"""bench.py as the driver runs it (a child process, one JSON line on stdout): the N = 1 contract fields, and the N > 1 code path --
RCCL process group, all-gather of the queries, per-shard distance GEMM, all_reduce(MIN) of the packed keys -- exercised on one
GPU with CMDIAD_FORCE_DIST=1 (a world of one rank goes through the same collectives)."""
import json
import os
import subprocess
import sys

import pytest

import time

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
slow = pytest.mark.slow      # still selected by `-m gpu`; each prints its duration (the GPU suite has a 20-minute budget on the driver)


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *args],
                         capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    print(f"[bench.py {' '.join(args) or '(default legs)'} {extra_env or ''}: {time.perf_counter() - t0:.0f} s]")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines          # exactly ONE line on stdout (RCCL's banner included: it goes to stderr)
    return json.loads(lines[0])


@pytest.mark.rehearsal
def test_bench_single_gpu_line():
    d = _run({}, "--no-extras")
    assert d["metric"].startswith("images/sec") and d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["value"] > 100 and abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.2 < r["frac"] < 1.0 and "traffic" in r and 0.1 < r["frac_in_pipeline"] <= r["frac"] * 1.05
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"


@slow
def test_bench_default_line_carries_the_secondary_legs():
    """The driver's command (no flags beyond steps): besides the headline, configs[2] (distillation training step), the conv head's
    training step (row f4), the variable-N leg, the MTFI step (the metric's "distill" term) and the class loop
    ride in the same JSON line; no leg carries an error."""
    d = _run({})
    assert not [k for k, v in d.items() if isinstance(v, dict) and ("error" in v or "skipped" in v)], d
    # the headline contract (what test_bench_single_gpu_line checks on a --no-extras run)
    assert d["metric"].startswith("images/sec") and d["unit"] == "images/s" and d["n_gpus"] == 1
    assert d["value"] > 100 and abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.2 < r["frac"] < 1.0 and "traffic" in r and 0.1 < r["frac_in_pipeline"] <= r["frac"] * 1.05
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    ms = d["mtfi_step"]
    assert ms["value"] > 100 and ms["ms_per_step"] > 0 and ms["query_rows"]["libraries"] == 2 and ms["hallucination_mlp"]["TFLOPs"] > 100
    t, ct = d["train_step"], d["conv_head_train_step"]
    assert t["ms_per_step"] > 0 and t["loss_last_timed"] < t["loss_first_timed"]
    assert ct["batch"] == 8 and ct["achieved_TFLOPs"] > 100 and ct["loss_last"] < ct["loss_first"]
    assert d["var_n"]["value"] > 100 and d["h2d_inclusive"]["value"] > 100 and d["every_row_searched"]["value"] > 100
    assert d.get("cpu_baseline") is None or d["cpu_baseline"]["value"] > 0
    assert len(d["mtfi_classes"]["per_class"]) == 10


@slow
@pytest.mark.rehearsal
def test_bench_distributed_path_on_one_gpu():
    d = _run({"CMDIAD_FORCE_DIST": "1"})
    assert d["rccl_ranks"] == 1 and d["world"] == 1 and d["ranks"][0]["rank"] == 0 and "device" in d["ranks"][0]
    assert not [k for k, v in d.items() if isinstance(v, dict) and ("error" in v or "skipped" in v)], d
    s = d["sharded_search"]
    assert s["rccl_ranks"] == 1 and s["backend"] == "nccl"
    assert d["value"] > 100
    # compact-then-gather: the exchange carries the live rows only, and the line says what moved and where the time went
    c = s["classes"][0]
    assert c["cls"] == "bagel" and c["rows"] == 76518 and c["rows_this_rank"] == 76518
    assert len(c["live_rows_per_rank"]) == 1 and 0.5 * 100352 < c["live_rows_per_rank"][0] < 0.7 * 100352
    assert c["gathered_rows_per_rank"] % 256 == 0 and c["live_rows_per_rank"][0] <= c["gathered_rows_per_rank"] < 100352
    assert c["gather_MB_received_per_rank"] == 0.0 and c["gather_MB_received_without_compaction"] == 0.0      # a world of one
    assert set(c["serial_ms_rank0"]) == {"dedup_and_gather", "gemm", "reduce_and_expand"} and c["serial_ms_rank0"]["gemm"] > 1.0
    assert c["gemm_tflops_rank0"] > 500 and c["ms_per_search"] > 0
    m = d["mtfi_classes"]
    assert m["world"] == 1 and len(m["per_class"]) == 10 and m["assignment"][0][0] == "peach" and 0.0 <= m["mean"]["image_rocauc"] <= 1.0


def test_bench_row_sharded_pipeline_on_one_gpu():
    """`--bank sharded` with CMDIAD_FORCE_DIST=1: the row-sharded search INSIDE the timed pipeline (engine.sharded_min_keys in
    BatchPredictor.search: local de-duplication, counts exchange, all-gather of the live rows, per-shard GEMM, all_reduce(MIN),
    expansion) through RCCL with a world of one rank; the searched-row count says the compaction happened before the exchange."""
    d = _run({"CMDIAD_FORCE_DIST": "1"}, "--no-extras", "--bank", "sharded")
    assert d["config"]["bank"].startswith("row-sharded") and d["value"] > 100
    rows = d["config"]["xyz_query_rows"]
    assert rows["dedup"] and 0.5 * rows["per_step"] < rows["searched_per_step"] < 0.7 * rows["per_step"]
    assert 0.2 < d["roofline"]["frac"] < 1.0
