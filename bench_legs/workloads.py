"""Other workloads through the same pipelined predictor: variable point counts (`var_n`), the MTFI per-GPU step (`mtfi_step`,
the metric's "distill" term) and the class loop of configs[4] (`mtfi_classes`)."""
import os
import sys
import time

from .common import BATCH, DEFECT_SEVERITY, N_POINTS, PEAK_BF16_TFLOPS, XYZ_ROWS, make_batches, run_steps
from .roofline import isolated_xyz_search_ms

def var_n_leg(st, dev, steps=16, warm=8):
    """SURVEY 8(d) var-N regime: every cloud keeps a different share of the image -- foreground 35 ... 65 % of the 224 x 224 pixels
    (N ~ 17.5 k ... 32.6 k points) -- instead of the fixed 24 576 points of the headline batches: the same predictor, ragged
    point counts inside a batch of 32 (padded to the largest, per-sample lengths on the device), and a DIFFERENT share of
    repeated background rows in front of the xyz search."""
    import numpy as np
    import torch
    from cmdiad_amd.predictor import BatchPredictor, EventTimer
    from cmdiad_amd.synth import synth_cloud, synth_rgb
    rs = np.random.RandomState(8)
    batches, n_pts = [], []
    for j in range(2):
        fr = (0.35 + 0.30 * rs.rand(BATCH)) / 0.85     # synth_cloud's ellipse covers 0.85 x frac of the image
        pcs = torch.cat([synth_cloud(7000 + j * BATCH + i, float(fr[i])) for i in range(BATCH)])
        n_pts += [int((pcs[i] != 0).all(0).sum()) for i in range(BATCH)]
        rgb = torch.cat([synth_rgb(7000 + j * BATCH + i) for i in range(BATCH)]) if st["workload"] == "dino_pointmae" else None
        batches.append((rgb.to(dev) if rgb is not None else None, pcs.to(dev)))
    n_max = (max(n_pts) + 255) // 256 * 256
    timers = {"xyz": EventTimer(), "rgb": EventTimer()}
    pred = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=BATCH,
                          n_max=n_max, workload=st["workload"], halluc=st["halluc"], group=None,
                          use_graph=os.environ.get("CMDIAD_GRAPH", "1") != "0", timers=timers)
    first = run_steps(pred, batches, warm)
    for t in timers.values():
        t.pairs.clear()
    torch.cuda.synchronize()
    pred.live_rows.zero_()
    pred.xyz_searches = 0
    t0 = time.perf_counter()
    run_steps(pred, batches, steps, first)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    live = float(pred.live_rows.item()) / max(pred.xyz_searches, 1)
    l2_ms = isolated_xyz_search_ms(pred) or timers["xyz"].mean_ms()    # the last step's launch again, alone
    rows = st["bank_xyz"].shard_rows
    return dict(what="var-N regime (SURVEY 8d): foreground 35-65 % of the image per cloud, ragged point counts inside the batch of 32",
                value=round(BATCH * steps / dt, 2), unit="images/s per GPU", ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                points_per_cloud=dict(min=min(n_pts), mean=round(sum(n_pts) / len(n_pts), 1), max=max(n_pts), padded_to=n_max),
                xyz_query_rows=dict(per_step=BATCH * 3136, searched_per_step=round(live, 1)),
                xyz_search_ms=round(l2_ms, 3), xyz_search_TFLOPs=round(2.0 * live * rows * 768 / (l2_ms * 1e-3) / 1e12, 1))


def mtfi_step_leg(st, dev, steps=12, warm=8):
    """The metric's "distill" term, driver-timed: the per-GPU step of configs[4] -- MTFI feature-to-feature predict with main
    modality xyz (RGBorXYZWithOneHallucination.predict, multiple_features.py:474-573) at batch 32 in steady state: Point-MAE
    extraction -> xyz patches -> hallucinated rgb features (the distillation network's xyz -> rgb direction,
    hallucination_network.py:34-45) -> two library searches (xyz and hallucinated-feature library, 76 518 x 768 each) -> scoring
    tail.  Same engine, xyz library and inputs as the headline (`python bench.py --workload mtfi` times this step as `value`);
    outputs compared step to step.  The hallucination MLP is then timed alone on the live rows of the last step."""
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import runtime
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.predictor import BatchPredictor, EventTimer
    from cmdiad_amd.synth import synth_bank
    torch.manual_seed(0)
    bank_second = eng.Bank(synth_bank(XYZ_ROWS, 768, 4323).to(dev))
    halluc = runtime.PackedHallucination(HallucinationCrossModalityNetwork(None, 768, 768).state_dict(), device=dev)
    timers = {"xyz": EventTimer(), "rgb": EventTimer()}
    pred = BatchPredictor(st["engine"], st["bank_xyz"], bank_second, st["stats"], st["det"], st["seg"], batch=BATCH, n_max=N_POINTS,
                          workload="mtfi", halluc=halluc, group=None, use_graph=os.environ.get("CMDIAD_GRAPH", "1") != "0", timers=timers)
    batches = [(None, p.to(dev)) for _, p in make_batches(0, "mtfi")]
    first = run_steps(pred, batches, warm)
    for t in timers.values():
        t.pairs.clear()
    torch.cuda.synchronize()
    pred.live_rows.zero_()
    pred.xyz_searches = 0
    t0 = time.perf_counter()
    run_steps(pred, batches, steps, first)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    live = float(pred.live_rows.item()) / max(pred.xyz_searches, 1)
    assert len({first[j][0].tobytes() for j in first}) == len(first), "the rotated batches must give distinct outputs"
    # the hallucination MLP alone: LN + 768 -> 1920 -> 1920 -> 768 (GELU after each) on the rows the step ran it on
    rows = int(round(live))
    x = torch.randn(rows, 768, device=dev)
    ev = EventTimer()
    for _ in range(6):
        with ev:
            halluc.generate(x, "xyz")
    torch.cuda.synchronize()
    h_ms = ev.mean_ms(skip=1)
    h_flops = 2.0 * rows * (768 * 1920 + 1920 * 1920 + 1920 * 768)
    lib = st["bank_xyz"].shard_rows
    return dict(what="configs[4] per-GPU step: MTFI FtoF predict, main modality xyz (Point-MAE extraction + hallucinated rgb features + "
                     "kNN score against the xyz and the hallucinated-feature libraries, 76518 x 768 each), batch 32, steady state, "
                     "inputs resident in HBM, outputs compared step to step",
                value=round(BATCH / dt, 2), unit="images/s per GPU", ms_per_step=round(dt * 1e3, 3), steps=steps, warmup=warm,
                query_rows=dict(per_step_per_library=BATCH * 3136, searched_per_step_per_library=round(live, 1), libraries=2,
                                note="both searches share the xyz patches' row plan: a patch without a foreground pixel is one repeated "
                                     "row in the xyz features AND in the features hallucinated from them"),
                search_ms_in_pipeline=dict(xyz=round(timers["xyz"].mean_ms(), 3), hallucinated=round(timers["rgb"].mean_ms(), 3)),
                search_TFLOPs_in_pipeline=round(2.0 * 2.0 * live * lib * 768 / ((timers["xyz"].mean_ms() + timers["rgb"].mean_ms()) * 1e-3) / 1e12, 1),
                hallucination_mlp=dict(rows=rows, ms_alone=round(h_ms, 3), GFLOP=round(h_flops / 1e9, 1),
                                       TFLOPs=round(h_flops / (h_ms * 1e-3) / 1e12, 1),
                                       frac_of_mfma_peak=round(h_flops / (h_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)))


def mtfi_classes(dev, group, rank, world, classes="all", scale=0.05, n_test=20, f_coreset=0.1):
    """configs[4] as a config: the reference's class loop (main.py:22-37 -> cmdiad_runner.CMDIAD.fit / evaluate) for the MTFI
    feature-to-feature method (RGBorXYZWithOneHallucination, main modality xyz) over synthetic stand-ins of the ten MVTec
    3D-AD classes, the classes dealt to the ranks by LPT (cmdiad_amd.evaluate), each class start to finish on its rank --
    memory bank, greedy coreset of both libraries, late-fusion bank, the two one-class SVMs, predict, I-/P-AUROC + AU-PRO --
    and ONE all_gather_object of the metric dictionaries at the end.  Train-set sizes are the MVTec counts times `scale`
    (so the relative class costs, hence the assignment and its imbalance, are those of the real data set)."""
    import torch
    from cmdiad_amd import evaluate as ev
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import sharpen_pointmae
    os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")    # synthetic weights: no checkpoints offline
    torch.manual_seed(0)
    weights = ({k: v.detach() for k, v in VisionTransformer().state_dict().items()},
               sharpen_pointmae({k: v.detach() for k, v in PointTransformer().state_dict().items()}),
               {k: v.detach() for k, v in HallucinationCrossModalityNetwork(None, 768, 768).state_dict().items()})
    names = "all" if classes == "all" else [c for c in classes.split(",") if c]
    data = ev.synthetic_mvtec3d(names, scale=scale, n_test=n_test, severity=DEFECT_SEVERITY)
    a = ev.mtfi_args(f_coreset=f_coreset)
    import contextlib
    import warnings
    if group is not None:
        import torch.distributed as td
        td.barrier(group)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):   # the drop-in prints the reference's progress lines
        warnings.simplefilter("ignore")
        res = ev.evaluate_classes(a, data, group=group, weights=weights)
    torch.cuda.synchronize()
    if group is not None:
        td.barrier(group)
    wall = time.perf_counter() - t0
    pc = res["per_class"]
    n_images = sum(v["n_test"] for v in pc.values())
    pred_s = [0.0] * world
    for v in pc.values():
        pred_s[v["rank"]] += v["seconds"]["predict"]
    return dict(what=f"class-sharded MTFI FtoF evaluation (fit -> predict -> metrics per class, {len(pc)} synthetic classes with "
                     f"MVTec 3D-AD train counts x {scale}, {n_test} test images each, f_coreset {f_coreset}), LPT over {world} rank(s), "
                     "metrics gathered with one all_gather_object",
                method=res["method"], world=world, assignment=res["assignment"], rank_seconds=res["rank_seconds"],
                wall_s=round(wall, 3), test_images=n_images,
                predict_images_per_s=round(n_images / max(max(pred_s), 1e-9), 1),
                job_images_per_s=round(n_images / wall, 2),
                per_class={c: {**{m: round(v[m], 4) for m in ev.METRICS}, "rank": v["rank"], "n_train": v["n_train"],
                               "n_test": v["n_test"], "seconds": v["seconds"], "library_rows": v["library_rows"]} for c, v in pc.items()},
                mean={m: res["table"][m]["Mean"] for m in ev.METRICS}, defect_severity=DEFECT_SEVERITY,
                host_fit_overlapped=os.environ.get("CMDIAD_EVAL_OVERLAP", "1") != "0" and world < len(pc))



# --------------------------------------------------------------------------------------------------------- main
