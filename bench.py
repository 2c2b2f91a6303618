#!/usr/bin/env python3
"""Headline benchmark: images/sec of the end-to-end `predict` hot path (extract + kNN score) on
synthetic MVTec-3D-shaped inputs -- BASELINE.json configs[1]: DINO ViT-B/8 + Point-MAE, 224x224 RGB +
1024-group point clouds, bf16 MFMA, batch 32 per GPU, 'bagel'-sized patch libraries
(xyz 76 518 x 768, rgb 19 129 x 768).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child before anything touches the GPU and
relays rank 0's JSON line (the driver's own torch.distributed.run launch works the same way: RANK / WORLD_SIZE are
read from the environment).

One step = one batch of 32 images per GPU through cmdiad_amd.predictor.BatchPredictor (= engine.predict_batch):
unorganise -> ViT-B/8 -> FPS -> kNN-group -> Point-MAE encoder + transformer -> 3-NN interpolation + 3x3/adaptive
pooling (fused) -> normalise -> distance GEMM with running (min, argmin) against both libraries -> exact re-score ->
re-weighting scan -> bilinear 224x224 maps -> 8-bit Gaussian blur (Pillow's arithmetic, bit-exact, on device) ->
lambda weights and the two linear one-class-SVM scores (models fitted on the host, scored on device) -> D2H of the
final image scores and pixel maps.  FOUR distinct batches are resident in HBM before the timed region and are rotated
step after step (D2D into the predictor's input buffers on its copy stream); every step's outputs are compared with the
first outputs of the same batch index.  The PCIe-inclusive rate (the same four batches in pinned host memory, H2D inside
the loop) is measured separately and reported as `h2d_inclusive` -- it is never `value`.

N > 1 (weak scaling): every rank scores its own batches against its own full copy of the libraries -- images are
independent, so `value` has no collective on the data path, only the barrier and the max-over-ranks of the timing.  The
SAME run then measures the north-star split of configs[3] and reports it as `sharded_search`: library rows sharded over
the ranks, all-gather of every rank's 16-bit queries, per-shard distance GEMM, ONE integer-MIN all-reduce of packed
(distance, global row) keys over RCCL -- for the bagel library and for all ten MVTec-3D class sizes (65 856 ... 113 209
rows).  CMDIAD_FORCE_DIST=1 exercises that path with a world of one rank on a single GPU.

Prints ONE JSON line (rank 0) with the fields of the bench contract plus `roofline` (dominant kernel: the xyz-library
distance GEMM, MFMA-bound; duration from HIP events inside the timed region) and, at N = 1, `cpu_baseline` (the CPU
oracle pipeline timed on a bounded sample on this box's host cores: best thread count, the reference's default 6
threads, and all cores) and `dropin_b1` (the B = 1 drop-in protocol the reference's main.py drives).

Half of the xyz query rows of a step repeat ONE row (the patches of the 56 x 56 grid without a foreground pixel); the search takes
that row once (csrc/dedup.hip, outputs bit-identical).  `roofline.achieved` counts the FLOPs executed, `config.xyz_query_rows` says
how many rows that was, and `every_row_searched` times the same steps with all rows searched as the reference's cdist does
(CMDIAD_DEDUP=0), comparing every output with the default run's.

Layout: this file holds main() only -- argument parsing, the process group, THE TIMED REGION and the JSON line.  Every leg lives in
bench_legs/ (runner: launcher + LegRunner + exit codes; common: resident state, batches, step loop; roofline; pipeline: h2d_inclusive /
every_row_searched; sharded; workloads; training; dropin; cpu: the only importer of oracle/).
"""
import argparse
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from bench_legs.common import (BATCH, CLASS_TRAIN, N_POINTS, PEAK_BF16_TFLOPS, PEAK_HBM_GBS, ROTATE, RGB_ROWS, XYZ_ROWS,  # noqa: E402,F401
                               build_state, class_rows, make_batches, run_steps)
from bench_legs.cpu import cpu_baseline  # noqa: E402
from bench_legs.dropin import dropin_b1  # noqa: E402
from bench_legs.pipeline import every_row_leg, h2d_leg  # noqa: E402
from bench_legs.roofline import isolated_xyz_search_ms, profiled_traffic  # noqa: E402,F401
from bench_legs.runner import (EXIT_OUT_OF_STEP, LegRunner, _free_port, emit_line, launch, leave_out_of_step,  # noqa: E402,F401
                               selftest_launch)
from bench_legs.sharded import fake_world_leg, sharded_search  # noqa: E402
from bench_legs.training import conv_head_train_leg, train_step_leg  # noqa: E402
from bench_legs.workloads import mtfi_classes, mtfi_step_leg, var_n_leg  # noqa: E402


def _traffic_with_ratio(t, bytes_alg):
    """+ how many times the launch's algorithmic bytes (operands read once, keys written) crossed the fabric: the wasted-traffic
    ratio of the decomposition (the library is re-streamed once per group of four query tiles; most of it is served by the L2 /
    Infinity Cache side of the fabric counters, `l2_hit` in traffic_note)."""
    if t.get("traffic"):
        t["traffic_over_algorithmic"] = round(t["traffic"] / bytes_alg, 1)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (h2d_inclusive, dropin_b1, sharded_search)")
    ap.add_argument("--cpu-images", type=int, default=10)
    ap.add_argument("--workload", choices=("dino_pointmae", "mtfi"), default="dino_pointmae",
                    help="dino_pointmae = BASELINE configs[1] (both modalities extracted, the headline workload); mtfi = the "
                         "per-GPU work of configs[4]: Point-MAE extraction + hallucinated rgb features + two library searches")
    ap.add_argument("--bank", choices=("replicated", "sharded"), default=os.environ.get("CMDIAD_BANK", "replicated"),
                    help="what `value` measures at N > 1: 'replicated' = every rank scores its own images against a full copy of "
                         "the libraries (no data-path collective); 'sharded' = the row-sharded search inside the pipeline.  The "
                         "row-sharded search is reported as `sharded_search` either way")
    ap.add_argument("--classes", default="all", help="'all' (ten MVTec-3D class sizes) or a comma list, for `sharded_search`")
    ap.add_argument("--evaluate", action="store_true",
                    help="configs[4] as a config instead of the timed predict loop: the reference's class loop (fit -> predict -> "
                         "I-/P-AUROC, AU-PRO per class) for the MTFI FtoF method over synthetic MVTec-3D-sized classes, classes "
                         "dealt to the ranks by LPT, metric dictionaries gathered at the end (cmdiad_amd/evaluate.py)")
    ap.add_argument("--class-scale", type=float, default=0.05, help="train images per class = MVTec 3D-AD count x this (1.0 = full size)")
    ap.add_argument("--class-test", type=int, default=20, help="test images per synthetic class (3 of every 10 anomalous)")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args.gpus, sys.argv[1:]))        # before ANY GPU call in this process
    if args.selftest_launch:
        sys.exit(selftest_launch())

    import numpy as np
    import torch
    from cmdiad_amd.predictor import BatchPredictor, EventTimer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (cmdiad_amd has no CPU fallback)")
    # Rehearsal of the N > 1 line on a ONE-GPU box (tests/test_gpu_world2.py): every rank uses device 0 and the collectives go
    # over gloo, because RCCL refuses two ranks on one device.  The code path is the driver's N > 1 path; the numbers are not.
    rehearsal = world > 1 and os.environ.get("CMDIAD_BENCH_ONE_DEVICE", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    force_dist = os.environ.get("CMDIAD_FORCE_DIST", "0") == "1"  # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as td
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
        # RCCL prints its version banner to STDOUT when the first communicator comes up: stdout carries exactly one JSON line
        # (the contract), so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            # beyond every leg budget of LegRunner: its watchdog ends a hung leg with the line printed; torch's would SIGABRT the job
            if rehearsal:
                td.init_process_group("gloo", timeout=datetime.timedelta(minutes=45))
            else:
                td.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=45))
            group = td.group.WORLD
            warm = torch.ones(1, device=dev)
            td.all_reduce(warm, group=group)
            torch.cuda.synchronize()
            rccl_ranks_seen = int(warm.item())             # what RCCL itself summed over: one per rank that joined
            census = [None] * td.get_world_size()
            td.all_gather_object(census, dict(rank=rank, local_rank=local, device=torch.cuda.get_device_name(local),
                                              pci_bus_id=getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None),
                                              host=socket.gethostname(), pid=os.getpid()))
        finally:
            sys.stdout.flush()
            import ctypes
            ctypes.CDLL(None).fflush(None)   # the banner sits in the C library's stdout buffer (a pipe is fully buffered)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    sharded = group is not None and args.bank == "sharded"

    if args.evaluate:
        leg = mtfi_classes(dev, group, rank, world, args.classes, args.class_scale, args.class_test)
        if group is not None:
            td.barrier()
            td.destroy_process_group()
        if rank == 0:
            line = {"metric": "images/sec end-to-end (extract+distill+kNN score)", "value": leg["predict_images_per_s"],
                    "unit": "images/s", "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": round(leg["wall_s"] * 1e3, 1),
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": "configs[4]: full MTFI FtoF pipeline per class (fit: memory bank, coreset, late-fusion bank, "
                                           "one-class SVMs; predict; I-/P-AUROC + AU-PRO), per-GPU class sharding (LPT)",
                               "value_is": "test images / slowest rank's predict seconds; job_images_per_s divides by the whole "
                                           "fit + predict + metrics wall time", "class_scale": args.class_scale},
                    "mtfi_classes": leg}
            emit_line(line)
        return

    st = build_state(dev, args.workload)
    if sharded:  # the pipeline itself searches row shards
        from cmdiad_amd import engine as eng
        st["bank_xyz"] = eng.Bank(st["bank_xyz"].f32, rank, world)
        st["bank_second"] = eng.Bank(st["bank_second"].f32, rank, world)
    timers = {"xyz": EventTimer(), "rgb": EventTimer()}
    pred = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=BATCH,
                          n_max=N_POINTS, workload=args.workload, halluc=st["halluc"], group=group if sharded else None,
                          use_graph=os.environ.get("CMDIAD_GRAPH", "1") != "0", timers=timers)
    host_batches = make_batches(rank, args.workload, pinned=True)
    batches = [(r.to(dev) if r is not None else None, p.to(dev)) for r, p in host_batches]   # resident in HBM

    first = run_steps(pred, batches, args.warmup)
    for t in timers.values():
        t.pairs.clear()
    if group is not None:
        td.barrier()
    torch.cuda.synchronize()
    pred.live_rows.zero_()
    pred.xyz_searches = 0
    t0 = time.perf_counter()
    run_steps(pred, batches, args.steps, first)
    torch.cuda.synchronize()
    if group is not None:
        td.barrier()
    dt = time.perf_counter() - t0
    q_live = float(pred.live_rows.item()) / max(pred.xyz_searches, 1)   # query rows per xyz search after the exact row de-duplication
    l2_alone_ms = isolated_xyz_search_ms(pred)                          # after the timed region: the dominant kernel alone
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if group is not None:
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
    dt = float(tmax.item())
    distinct = len({first[j][0].tobytes() for j in first})
    assert distinct == len(first), "the rotated batches must give distinct outputs"

    # ---- the headline is complete here: everything below is a secondary leg and runs under LegRunner (fault-isolated, budgeted)
    out = None
    if rank == 0:
        images = BATCH * world * args.steps
        l2_pipe_ms = timers["xyz"].mean_ms()
        l2_ms = l2_alone_ms if l2_alone_ms else l2_pipe_ms
        q_total = BATCH * 3136 * (world if sharded else 1)
        rows = st["bank_xyz"].shard_rows
        # FLOPs of the launch as executed: the rows the kernel searched (patches without a foreground pixel repeat one row and are
        # searched once, csrc/dedup.hip; CMDIAD_DEDUP=0 searches all q_total rows as the reference's cdist does)
        flops = 2.0 * q_live * rows * 768
        achieved = flops / (l2_ms * 1e-3) / 1e12
        achieved_pipe = flops / (l2_pipe_ms * 1e-3) / 1e12 if l2_pipe_ms else None
        bytes_alg = (rows + q_live) * 768 * 2 + 12 * q_live
        out = {
            "metric": "images/sec end-to-end (extract+distill+kNN score)", "value": round(images / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: DINO ViT-B/8 + Point-MAE predict (DoubleRGBPointFeatures: both modalities are "
                                    "extracted, this method has no distillation step at test time; the `mtfi_step` leg times the "
                                    "method that has one), 224x224 RGB + "
                                    "24576-point clouds (1024 groups x 128), batch 32/GPU, bagel-sized banks "
                                    "(xyz 76518x768, rgb 19129x768)") if args.workload == "dino_pointmae" else
                                   ("configs[4] per-GPU work: MTFI FtoF predict (RGBorXYZWithOneHallucination, main modality "
                                    "xyz): Point-MAE extraction + hallucinated rgb features (distillation network) + kNN "
                                    "score against the xyz and the hallucinated-feature libraries (76518x768 each), "
                                    "24576-point clouds, batch 32/GPU"),
                       "batch_per_gpu": BATCH, "rotating_input_batches": ROTATE,
                       "bank": "row-sharded search + RCCL min-reduce" if sharded else ("replicated per rank, images sharded, no data-path collective" if world > 1 else "single"),
                       "xyz_query_rows": {"per_step": q_total, "searched_per_step": round(q_live, 1), "dedup": bool(pred.dedup),
                                          "note": "the 56x56 patch grid keeps a row for every patch; patches with no foreground pixel "
                                                  "(24576 of 50176 pixels are foreground, as in the reference's clouds) are one "
                                                  "repeated row, searched once with the key copied -- results identical to searching all"},
                       "hip_graphs": bool(pred.use_graph),
                       "search_operands": ("bf16" if st["bank_xyz"].bf16.dtype == torch.bfloat16 else "fp16") + " (fp32 accumulate, exact fp32 re-score of the winner)",
                       "weights": "seeded random init (no checkpoints offline)"},
            "roofline": {"kernel": "l2_min_pp3_kernel (xyz library distance GEMM + running min/argmin)", "bound": "mfma",
                         "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "frac_in_pipeline": round(achieved_pipe / PEAK_BF16_TFLOPS, 4) if achieved_pipe else None,
                         **_traffic_with_ratio(profiled_traffic(), bytes_alg),
                         "launch_ms": round(l2_ms, 3), "launch_ms_in_pipeline": round(l2_pipe_ms, 3),
                         "launch_ms_note": "launch_ms / frac: the step's launch repeated alone after the timed loop (HIP events, idle chip; median of ten) "
                                           "-- the regime of the stand-alone rocprofv3 row in profiles/r5_standalone.md; "
                                           "in_pipeline: the same launch inside the timed steps, on the second stream beside the next "
                                           "step's extraction (shares the CUs)",
                         "flops_per_launch": flops,
                         "hbm_secondary": {"algorithmic_bytes": bytes_alg,
                                           "achieved_GBs": round(bytes_alg / (l2_ms * 1e-3) / 1e9, 1),
                                           "frac_of_8TBs": round(bytes_alg / (l2_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}},
        }
        if group is not None:
            out["rccl_ranks"] = rccl_ranks_seen          # the sum RCCL's all_reduce returned over a tensor of ones
            out["world"] = world
            out["ranks"] = census
            out["backend"] = td.get_backend(group)
            if rehearsal:
                out["rehearsal"] = f"{world} ranks share device 0, collectives over gloo (CMDIAD_BENCH_ONE_DEVICE=1): a code-path check, not a measurement"
    store = None
    if group is not None:
        store = td.distributed_c10d._get_default_store()
    legs = LegRunner(out, rank, store, emit_line)
    budget = float(os.environ.get("CMDIAD_BENCH_LEG_BUDGET", "0")) or None   # one budget for every leg (tests); default: per leg

    only = [x for x in os.environ.get("CMDIAD_BENCH_LEGS", "").split(",") if x]   # A/B runs: only these secondary legs (default: all)

    def leg(name, fn, seconds, collective=False):
        if only and name not in only and name != "teardown":
            return None
        return legs.run(name, fn, budget or seconds, collective=collective)

    if not args.no_extras:
        n_h2d = max(8, min(args.steps, 12))

        # (with the row-sharded pipeline -- --bank sharded -- the H2D-fed steps are collective steps too)
        leg("h2d_inclusive", lambda: h2d_leg(pred, host_batches, 2 * n_h2d, first), 120, collective=sharded)   # (its first two steps are not staged ahead)
        if group is None and pred.dedup:
            leg("every_row_searched", lambda: every_row_leg(st, pred, batches, n_h2d, first, args.workload), 180)
        if group is not None:
            names = list(CLASS_TRAIN) if args.classes == "all" else [c for c in args.classes.split(",") if c]
            leg("sharded_search", lambda: sharded_search(dev, group, rank, world, [(c, class_rows(c)) for c in names]), 300, collective=True)
        if group is None:
            leg("fake_world", lambda: fake_world_leg(dev), 240)
            leg("var_n", lambda: var_n_leg(st, dev), 120)
            if args.workload == "dino_pointmae":
                leg("mtfi_step", lambda: mtfi_step_leg(st, dev), 180)
            leg("train_step", lambda: train_step_leg(dev), 120)
            leg("conv_head_train_step", lambda: conv_head_train_leg(dev), 120)
        # configs[4] as a config (bounded): the class loop with the classes dealt to the ranks, metrics gathered at the end
        leg("mtfi_classes", lambda: mtfi_classes(dev, group, rank, world, "all", args.class_scale, args.class_test), 600,
            collective=group is not None)

    torn_down = False
    if group is not None and legs.in_step:
        def teardown():
            td.barrier()
            td.destroy_process_group()
            legs.store = None            # (gone with the process group; nothing collective follows)
        torn_down = "error" not in (leg("teardown", lambda: teardown() or {"ok": True}, 60, collective=True) or {})
    if rank == 0:
        if world == 1 and not args.no_extras and args.workload == "dino_pointmae":
            del pred, batches
            torch.cuda.empty_cache()
            leg("dropin_b1", dropin_b1, 300)
        if world == 1 and not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline(args.cpu_images), 600)
        out.pop("teardown", None)
        emit_line(out)
    if group is not None and not torn_down:
        # some rank failed a leg: the others may sit in a collective that never completes -- no orderly teardown, non-zero exit
        leave_out_of_step(store, rank)


if __name__ == "__main__":
    main()
