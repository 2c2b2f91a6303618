#!/usr/bin/env python3
"""Headline benchmark: images/sec of the end-to-end `predict` hot path (extract + kNN score) on
synthetic MVTec-3D-shaped inputs -- BASELINE.json configs[1]: DINO ViT-B/8 + Point-MAE, 224x224 RGB +
1024-group point clouds, bf16 MFMA, batch 32 per GPU, 'bagel'-sized patch libraries
(xyz 76 518 x 768, rgb 19 129 x 768).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one batch of 32 images per GPU through: unorganise -> ViT-B/8 -> FPS -> kNN-group ->
Point-MAE encoder + transformer -> 3-NN interpolation + 3x3/adaptive pooling (fused) -> normalise ->
distance GEMM with running (min, argmin) against both libraries -> exact re-score -> re-weighting scan
-> bilinear 224x224 maps -> 8-bit Gaussian blur (Pillow's arithmetic, bit-exact, on device) -> lambda weights and
the two linear one-class-SVM scores (models fitted on the host, scored on device) -> D2H of the final image
scores and pixel maps.
Inputs are resident in HBM before the timed region.  With N > 1 every rank processes its own batch
(weak scaling; images are independent, so there is no collective on the data path -- only the barrier and the
max-over-ranks of the timing).  --bank sharded (or CMDIAD_BANK=sharded) switches the library SEARCH to row shards: all-gather of the
16-bit queries, per-shard distance GEMM, one integer-MIN all-reduce of packed keys over RCCL.

Prints ONE JSON line (rank 0) with the fields of the bench contract plus `roofline` (dominant kernel:
the xyz-library distance GEMM, MFMA-bound) and, at N = 1, `cpu_baseline` (the CPU oracle pipeline timed
on a bounded sample on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BATCH = 32
N_POINTS = 24576          # fixed-N regime of the batch-32 config
XYZ_ROWS, RGB_ROWS = 76518, 19129   # floor(0.1 * 244 * 3136), floor(0.1 * 244 * 784): 'bagel'
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def build_state(dev, rank, world, workload="dino_pointmae"):
    from cmdiad_amd import engine as eng
    from cmdiad_amd import runtime
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb
    torch.manual_seed(0)  # random-init weights of the named architectures (no checkpoints offline)
    vit = runtime.PackedViT(VisionTransformer().state_dict(), device=dev)
    pm = runtime.PackedPointMAE(PointTransformer().state_dict(), device=dev)
    e = eng.Engine(vit, pm)
    rgb = torch.cat([synth_rgb(rank * BATCH + i) for i in range(BATCH)]).to(dev)
    pcs = torch.cat([synth_cloud_fixed_n(1000 + rank * BATCH + i, N_POINTS) for i in range(BATCH)]).to(dev)
    bank_xyz = eng.Bank(synth_bank(XYZ_ROWS, 768, 4321).to(dev), rank, world)
    if workload == "mtfi":
        # MTFI feature-to-feature, main modality xyz (multiple_features.py:312-573): the rgb sensor is absent at test time;
        # its features are hallucinated from the xyz patches and scored against the library of hallucinated train features
        # (one row per 56 x 56 patch -> as many rows as the xyz library)
        bank_rgb = eng.Bank(synth_bank(XYZ_ROWS, 768, 4323).to(dev), rank, world)
        halluc = runtime.PackedHallucination(HallucinationCrossModalityNetwork(None, 768, 768).state_dict(), device=dev)
    else:
        bank_rgb = eng.Bank(synth_bank(RGB_ROWS, 768, 4322).to(dev), rank, world)
        halluc = None
    # scalar library statistics (cross-wired as the reference): synthetic banks are N(0,1)
    stats = dict(xyz_mean=0.0, xyz_std=1.0, rgb_mean=0.0, rgb_std=1.0)
    # late-fusion linear one-class SVMs fitted on synthetic score rows (host sklearn)
    from sklearn import linear_model
    rs = np.random.RandomState(0)
    det = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 2))
    seg = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 2))
    return dict(engine=e, rgb=rgb, pcs=pcs, bank_xyz=bank_xyz, bank_rgb=bank_rgb, stats=stats, det=det, seg=seg,
                halluc=halluc, workload=workload)


class Timer:
    """HIP-event bracket on torch's current stream (the stream every cmdiad kernel is launched on)."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record()
        return self

    def __exit__(self, *a):
        self.e1.record()
        self.pairs.append((self.e0, self.e1))

    def mean_ms(self, skip=0):
        v = [a.elapsed_time(b) for a, b in self.pairs[skip:]]
        return sum(v) / max(len(v), 1)


class Pipeline:
    """One step = stage1 (extract + queries; HIP graph) -> search (distance GEMMs, eager and bracketed by HIP
    events so `roofline` is measured live; the collectives of the sharded mode live here) -> stage2 (re-score,
    re-weighting, score maps; HIP graph) -> async D2H into a ring of pinned buffers.

    The two graphs remove the Python/ctypes launch path of ~300 small launches per step from the critical
    path (guide G9: the C ABI never synchronises or allocates, so capture is legal); set CMDIAD_GRAPH=0 to run
    everything eagerly.  With graphs, two buffer sets alternate and stage2 + D2H of step i run on a second stream beside
    stage1 of step i+1 (step()); main() asserts that every step of the run returns identical outputs."""

    def __init__(self, st, group, timers, use_graph=True, ring=3):
        self.st, self.group, self.timers = st, group, timers
        self.side = torch.cuda.Stream()
        self.post = torch.cuda.Stream()   # scoring tail of the previous step (see step())
        self.step_no = 0
        # host ring: the step's FINAL outputs (image score, pixel map), f64 as sklearn's score_samples returns them
        self.ring = [(torch.empty((BATCH, 1), dtype=torch.float64, pin_memory=True),
                      torch.empty((BATCH, 224 * 224), dtype=torch.float64, pin_memory=True)) for _ in range(ring)]
        self.slot = 0
        self.g1 = self.g2 = None
        self.use_graph = use_graph
        self.static = {}

    # ---- stage 1: everything up to the bf16 queries of both modalities
    def stage1(self):
        from cmdiad_amd import engine as eng
        from cmdiad_amd import ops
        st = self.st
        e, s = st["engine"], st["stats"]
        if st["workload"] == "mtfi":
            ex = e.extract(None, st["pcs"], want_rgb=False, n_max=N_POINTS)
            xyz_raw = e.xyz_patch(ex, 56)                                       # a9
            hall = st["halluc"].generate(xyz_raw, "xyz")                        # a15: hallucinated rgb features [B,3136,768]
            xyz_q = eng.normalize(xyz_raw, s["xyz_mean"], s["xyz_std"])         # a11
            rgb_q = eng.normalize(hall, s["rgb_mean"], s["rgb_std"])
        else:
            early = {}

            def rgb_branch(ex):
                # everything the rgb library needs depends on the ViT only: normalise, cast and SEARCH it here, beside the
                # rest of the point-cloud branch (the Point-MAE transformer leaves half of the chip's issue slots idle)
                rq = eng.normalize(e.rgb_patch(ex).contiguous(), s["rgb_mean"], s["rgb_std"])
                B, Q, D = rq.shape
                q16, _, qsq = ops.normalize_cast(rq.reshape(B * Q, D))
                early["rgb"] = (rq, q16, qsq)
                if self.group is None:
                    bank = st["bank_rgb"]
                    k = torch.full((B * Q,), -1, dtype=torch.int64, device=rq.device)
                    ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, k, bank.row_offset)
                    early["rgb_keys"] = k

            ex = e.extract(st["rgb"], st["pcs"], n_max=N_POINTS, side_stream=self.side, rgb_hook=rgb_branch)
            xyz_q = e.xyz_patch(ex, 56, s["xyz_mean"], 1.0 / s["xyz_std"])        # a9 + a11 fused
            B, Q, D = xyz_q.shape
            q16, _, qsq = ops.normalize_cast(xyz_q.reshape(B * Q, D))
            out = {"xyz": (xyz_q, q16, qsq), "rgb": early["rgb"]}
            if "rgb_keys" in early:
                out["rgb_keys"] = early["rgb_keys"]
            return out
        out = {}
        for name, q in (("xyz", xyz_q), ("rgb", rgb_q)):
            B, Q, D = q.shape
            q16, _, qsq = ops.normalize_cast(q.reshape(B * Q, D))
            out[name] = (q, q16, qsq)
        return out

    # ---- search: eager (HIP events around the distance GEMM; RCCL collectives when sharded)
    def search(self, qs, buf=0):
        from cmdiad_amd import engine as eng
        from cmdiad_amd import ops
        keys = {}
        for name, bank in (("xyz", self.st["bank_xyz"]), ("rgb", self.st["bank_rgb"])):
            if name == "rgb" and "rgb_keys" in qs:  # searched inside stage 1 already (beside the point-cloud branch)
                keys[name] = qs["rgb_keys"]
                continue
            q, q16, qsq = qs[name]
            B, Q, D = q.shape
            q_all, s_all = eng.gather_queries(q16, qsq, self.group)
            k = self.static.get(f"keys_{name}_{buf}")
            if k is None or k.shape[0] != q_all.shape[0]:
                k = self.static[f"keys_{name}_{buf}"] = torch.empty((q_all.shape[0],), dtype=torch.int64, device=q.device)
            k.fill_(-1)
            with self.timers[name]:
                ops.l2_min_keys(q_all, s_all, bank.bf16, bank.sqnorm, k, bank.row_offset)
            k = eng.merge_shard_keys(k, self.group)
            keys[name] = k[bank.rank * B * Q:(bank.rank + 1) * B * Q] if self.group is not None else k
        return keys

    # ---- stage 2: exact re-score, re-weighting, bilinear maps, 8-bit blur (a14), lambda weights + one-class SVMs (a19)
    def stage2(self, qs, keys, lambdas=(1.0, 1.0, 0.1, 0.1)):
        from cmdiad_amd import engine as eng
        from cmdiad_amd import ops
        st = self.st
        rx = eng.score_patches_from_keys(qs["xyz"][0], keys["xyz"].contiguous(), st["bank_xyz"], (56, 56))
        side = (56, 56) if st["workload"] == "mtfi" else (28, 28)
        rr = eng.score_patches_from_keys(qs["rgb"][0], keys["rgb"].contiguous(), st["bank_rgb"], side)
        s = torch.stack([rx["s"], rr["s"]], 1)                                   # [B,2]
        maps = torch.stack([rx["s_map_pre"], rr["s_map_pre"]], 1).contiguous()   # [B,2,224,224]
        B = maps.shape[0]
        blurred = ops.blur8_maps(maps.view(B * 2, 224, 224), 4.0).view(B, 2, 224 * 224)
        pix = ops.ocsvm_score_maps(blurred, (lambdas[1], lambdas[3]), st["seg"].coef_, st["seg"].offset_)
        img = ops.ocsvm_score_maps(s.view(B, 2, 1).contiguous(), (lambdas[0], lambdas[2]), st["det"].coef_, st["det"].offset_)
        return img, pix

    def _capture(self):
        qs = self.stage1()  # one eager pass first: module loading / attribute setting must not happen in capture
        self.stage2(qs, self.search(qs, 0))
        torch.cuda.synchronize()
        try:
            self.sets = []
            for s in range(2):  # two complete buffer sets: step i+1's extraction overlaps step i's scoring tail
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    qs = self.stage1()
                g1.replay()
                keys = self.search(qs, s)
                k = {n: v.contiguous() for n, v in keys.items()}
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    out = self.stage2(qs, k)
                torch.cuda.synchronize()
                self.sets.append(dict(g1=g1, g2=g2, qs=qs, k=k, out=out, done=None))
            self.g1 = self.sets[0]["g1"]
        except Exception as exc:  # capture is an optimisation, never a requirement
            print(f"[bench] HIP graph capture unavailable ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            self.g1 = None
            self.use_graph = False
            torch.cuda.synchronize()

    def step(self):
        if self.use_graph and self.g1 is None:
            self._capture()
        if self.use_graph:
            # Software pipeline across batches: the scoring tail of step i (re-score, re-weighting scans, maps, blur, one-class
            # SVMs, D2H: ~2 ms of small bandwidth-bound kernels) runs on a second stream beside the extraction of step i+1,
            # which leaves most of the chip idle while farthest-point sampling walks its chain.  Two buffer sets alternate; a
            # set is reused only after its own tail has finished (event wait below).
            st = self.sets[self.step_no & 1]
            self.step_no += 1
            cur = torch.cuda.current_stream()
            if st["done"] is not None:
                cur.wait_event(st["done"])
            st["g1"].replay()
            keys = self.search(st["qs"], self.sets.index(st))
            for n, k in keys.items():
                if k.data_ptr() != st["k"][n].data_ptr():
                    st["k"][n].copy_(k)
            self.post.wait_stream(cur)
            host_s, host_m = self.ring[self.slot]
            self.slot = (self.slot + 1) % len(self.ring)
            with torch.cuda.stream(self.post):
                st["g2"].replay()
                s_dev, maps_dev = st["out"]
                host_s.copy_(s_dev, non_blocking=True)
                host_m.copy_(maps_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            st["done"] = ev
            return host_s, host_m, ev
        qs = self.stage1()
        s_dev, maps_dev = self.stage2(qs, self.search(qs, 0))
        host_s, host_m = self.ring[self.slot]
        self.slot = (self.slot + 1) % len(self.ring)
        host_s.copy_(s_dev, non_blocking=True)
        host_m.copy_(maps_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host_s, host_m, ev


def collect(host_s, host_m, ev):
    """Waits for the step's D2H copy and takes the final scores out of the pinned ring slot."""
    ev.synchronize()
    return host_s.numpy().copy(), host_m.numpy().copy()


def cpu_baseline(n_images=3):
    """The CPU oracle pipeline (oracle/pipeline.py, kind 'port': the reference's own torch-CPU composition +
    the C restatement of FPS / kNN) on a bounded sample of the same workload, host cores of this box."""
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb
    from oracle.pipeline import CpuDoubleRGBPoint, CpuExtractor
    torch.manual_seed(0)
    sd_vit = {k: v.detach() for k, v in VisionTransformer().state_dict().items()}
    sd_pm = {k: v.detach() for k, v in PointTransformer().state_dict().items()}
    cpu = CpuDoubleRGBPoint(CpuExtractor(sd_vit, sd_pm))
    cpu.set_banks(synth_bank(XYZ_ROWS, 768, 4321), synth_bank(RGB_ROWS, 768, 4322), 0.0, 1.0, 0.0, 1.0)
    # thread count: the fastest of {all cores, 64, 32, 16, 6} measured on the 128-core MI355X host (0.18 / 0.34 / 0.45 / 0.47 /
    # 0.35 images/s; torch's intra-op pool oversubscribes badly beyond ~32 threads on these shapes; 6 = the reference's
    # default --cpu_core_num, main.py:149) -- capped at 32 so the reported baseline is the CPU's best, not its worst
    all_threads = torch.get_num_threads()
    cores = min(32, all_threads)
    torch.set_num_threads(cores)
    cpu.predict(synth_rgb(0), synth_cloud_fixed_n(1000, N_POINTS))  # warm-up (page in, MKL init)
    cpu.ex.timing.clear(); cpu.timing.clear()
    t0 = time.perf_counter()
    for i in range(n_images):
        cpu.predict(synth_rgb(1 + i), synth_cloud_fixed_n(1001 + i, N_POINTS))
    dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    stages = {k: round(v / n_images, 4) for k, v in {**cpu.ex.timing, **cpu.timing}.items()}
    return dict(value=round(n_images / dt, 4), unit="images/s", cores=cores, kind="port",
                sample=f"{n_images} images after 1 warm-up, B=1, fp32, torch {torch.__version__} CPU ({cores} intra-op threads: the "
                       f"fastest setting on this {all_threads}-thread host) + C oracle for FPS/kNN, same synthetic inputs and "
                       f"bagel-sized banks", seconds_per_image_by_stage=stages)


def measured_traffic(world, sharded):
    """Fabric-side bytes per launch of the xyz l2_min kernel from the committed rocprofv3 PMC passes
    (profiles/r1_pmc.json via tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included).
    PMC counters cannot be read inside this process, so the number is the separately profiled run of this same
    command and workload; None when that file is absent or the workload differs (sharded bank)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc.json")
    if world != 1 or sharded or not os.path.exists(path):
        return None
    grid = 0
    best = None
    for row in json.load(open(path)):
        if row["kernel"].startswith("l2_min_") and row["grid_threads"] > grid and "fetch_bytes" in row:
            grid, best = row["grid_threads"], row
    return None if best is None else int(best["fetch_bytes"] + best.get("write_bytes", 0.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=6)
    ap.add_argument("--workload", choices=("dino_pointmae", "mtfi"), default="dino_pointmae",
                    help="dino_pointmae = BASELINE configs[1] (both modalities extracted, the headline workload); mtfi = the "
                         "per-GPU work of configs[4]: Point-MAE extraction + hallucinated rgb features + two library searches")
    ap.add_argument("--bank", choices=("replicated", "sharded"), default=os.environ.get("CMDIAD_BANK", "replicated"),
                    help="N > 1: 'replicated' = every rank scores its own images against a full copy of the libraries (no "
                         "data-path collective); 'sharded' = row-sharded library search with RCCL all-gather + MIN all-reduce")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (cmdiad_amd has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    force_dist = os.environ.get("CMDIAD_FORCE_DIST", "0") == "1"  # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as td
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
        td.init_process_group("nccl", device_id=dev)
        group = td.group.WORLD
    # Images are independent (SURVEY 8e axis ii): by default every rank scores its own batch against its own full copy of
    # the libraries (353 MB of 288 GB) -- no collective on the data path.  CMDIAD_BANK=sharded selects the row-sharded
    # search (axis i: all-gather of the queries, per-shard distance GEMM, integer-MIN all-reduce of packed keys over RCCL),
    # the mode for libraries that do not fit one GPU.
    sharded = (world > 1 or force_dist) and args.bank == "sharded"

    st = build_state(dev, rank if sharded else 0, world if sharded else 1, args.workload)
    timers = {"xyz": Timer(), "rgb": Timer()}
    g = group if sharded else None
    pipe = Pipeline(st, g, timers, use_graph=os.environ.get("CMDIAD_GRAPH", "1") != "0")

    def run(n):
        pending, out = [], []
        for _ in range(n):
            if len(pending) >= 2:  # pinned ring of 3: the slot reused next must have been consumed
                out.append(collect(*pending.pop(0)))
            pending.append(pipe.step())
        out.extend(collect(*p) for p in pending)
        return out

    run(args.warmup)
    for t in timers.values():
        t.pairs.clear()
    if group is not None:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(args.steps)
    torch.cuda.synchronize()
    if group is not None:
        td.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if group is not None:
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
    dt = float(tmax.item())
    assert all(np.isfinite(r[0]).all() and np.isfinite(r[1]).all() for r in res)
    # every step scores the same synthetic batch: identical outputs step after step (also guards the two-set pipelining)
    assert all(np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) for r in res), "steps disagree"

    if rank == 0:
        images = BATCH * world * args.steps
        l2_ms = timers["xyz"].mean_ms()
        q_total = BATCH * 3136 * (world if sharded else 1)
        rows = st["bank_xyz"].bf16.shape[0]
        flops = 2.0 * q_total * rows * 768
        achieved = flops / (l2_ms * 1e-3) / 1e12
        bytes_alg = (rows + q_total) * 768 * 2 + 12 * q_total
        out = {
            "metric": "images/sec end-to-end (extract+distill+kNN score)", "value": round(images / dt, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: DINO ViT-B/8 + Point-MAE predict (DoubleRGBPointFeatures: both modalities are "
                                    "extracted, this method has no distillation step at test time), 224x224 RGB + "
                                    "24576-point clouds (1024 groups x 128), batch 32/GPU, bagel-sized banks "
                                    "(xyz 76518x768, rgb 19129x768)") if args.workload == "dino_pointmae" else
                                   ("configs[4] per-GPU work: MTFI FtoF predict (RGBorXYZWithOneHallucination, main modality "
                                    "xyz): Point-MAE extraction + hallucinated rgb features (distillation network) + kNN "
                                    "score against the xyz and the hallucinated-feature libraries (76518x768 each), "
                                    "24576-point clouds, batch 32/GPU"),
                       "batch_per_gpu": BATCH, "bank": "row-sharded search + RCCL min-reduce" if sharded else ("replicated per rank, images sharded, no data-path collective" if world > 1 else "single"),
                       "hip_graphs": bool(pipe.use_graph), "search_operands": "fp16 (fp32 accumulate, exact fp32 re-score)",
                       "weights": "seeded random init (no checkpoints offline)"},
            "roofline": {"kernel": "l2_min_pp3_kernel (xyz library distance GEMM + running min/argmin)", "bound": "mfma",
                         "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": measured_traffic(world, sharded),
                         "launch_ms": round(l2_ms, 3), "flops_per_launch": flops,
                         "hbm_secondary": {"algorithmic_bytes": bytes_alg,
                                           "achieved_GBs": round(bytes_alg / (l2_ms * 1e-3) / 1e9, 1),
                                           "frac_of_8TBs": round(bytes_alg / (l2_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_images)
    if group is not None:
        td.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to C stdout, which is flushed at exit -- i.e. AFTER a Python print: push it out
        # first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
