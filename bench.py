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
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BATCH = 32
N_POINTS = 24576          # fixed-N regime of the batch-32 config
XYZ_ROWS, RGB_ROWS = 76518, 19129   # floor(0.1 * 244 * 3136), floor(0.1 * 244 * 784): 'bagel'
# MVTec 3D-AD train-set sizes [external counts, SURVEY 8d]: bank rows = floor(0.1 * n_train * 3136)
CLASS_TRAIN = {"bagel": 244, "cable_gland": 223, "carrot": 286, "cookie": 210, "dowel": 288, "foam": 236, "peach": 361,
               "potato": 300, "rope": 298, "tire": 210}
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 / fp16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
ROTATE = 4                          # distinct input batches rotated through the timed region
DEFECT_SEVERITY = float(os.environ.get("CMDIAD_DEFECT_SEVERITY", "0.22"))   # synthetic defects of the class loop: hard enough that I-AUROC is not saturated (synth.SyntheticClass)


def class_rows(name):
    return int(0.1 * CLASS_TRAIN[name] * 3136)


# --------------------------------------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(n, argv):
    """Parent of an N-rank run: nothing here may touch the GPU (a process that has initialised HIP must not exec or be
    replaced, and the children need the devices).  Children inherit stderr; rank 0's JSON line is the last stdout line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
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
    loop has already earned are printed, the leg carries {"error": ...} and the job ends with exit code 0.

    * An exception is caught, recorded in the leg and announced to the other ranks through the process group's key-value store (no
      collective: the ranks are no longer in step); collective legs that have not started yet are skipped everywhere.
    * A wall-clock budget per leg, kept by a watchdog thread on every rank: when it runs out, rank 0 prints the line as it stands
      (the running leg marked as timed out) and every rank leaves with os._exit(0) -- a process that has touched the GPU exits, it
      never re-executes anything.  (torch's own NCCL watchdog would abort the whole process group with SIGABRT instead: its
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
            os._exit(0)

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
        os._exit(0)          # the other ranks may sit in a collective that will never complete: no orderly teardown
    return 0 if res.get("merge_ok") else 1


# --------------------------------------------------------------------------------------------------------- state
def build_state(dev, workload="dino_pointmae"):
    import numpy as np
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import runtime
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import synth_bank
    torch.manual_seed(0)  # random-init weights of the named architectures (no checkpoints offline)
    vit = runtime.PackedViT(VisionTransformer().state_dict(), device=dev)
    pm = runtime.PackedPointMAE(PointTransformer().state_dict(), device=dev)
    e = eng.Engine(vit, pm)
    bank_xyz = eng.Bank(synth_bank(XYZ_ROWS, 768, 4321).to(dev))
    if workload == "mtfi":
        # MTFI feature-to-feature, main modality xyz (multiple_features.py:312-573): the rgb sensor is absent at test time;
        # its features are hallucinated from the xyz patches and scored against the library of hallucinated train features
        # (one row per 56 x 56 patch -> as many rows as the xyz library)
        bank_second = eng.Bank(synth_bank(XYZ_ROWS, 768, 4323).to(dev))
        halluc = runtime.PackedHallucination(HallucinationCrossModalityNetwork(None, 768, 768).state_dict(), device=dev)
    else:
        bank_second = eng.Bank(synth_bank(RGB_ROWS, 768, 4322).to(dev))
        halluc = None
    # scalar library statistics (cross-wired as the reference): synthetic banks are N(0,1)
    stats = dict(xyz_mean=0.0, xyz_std=1.0, rgb_mean=0.0, rgb_std=1.0)
    # late-fusion linear one-class SVMs fitted on synthetic score rows (host sklearn)
    from sklearn import linear_model
    rs = np.random.RandomState(0)
    det = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 2))
    seg = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 2))
    return dict(engine=e, bank_xyz=bank_xyz, bank_second=bank_second, stats=stats, det=det, seg=seg, halluc=halluc,
                workload=workload)


def make_batches(rank, workload, pinned=False):
    """ROTATE distinct batches of BATCH synthetic samples (host tensors; pinned for the H2D-inclusive measurement)."""
    import torch
    from cmdiad_amd.synth import synth_cloud_fixed_n, synth_rgb
    out = []
    for j in range(ROTATE):
        base = (rank * ROTATE + j) * BATCH
        rgb = torch.cat([synth_rgb(base + i) for i in range(BATCH)]) if workload == "dino_pointmae" else None
        pcs = torch.cat([synth_cloud_fixed_n(1000 + base + i, N_POINTS) for i in range(BATCH)])
        if pinned:
            rgb, pcs = (rgb.pin_memory() if rgb is not None else None), pcs.pin_memory()
        out.append((rgb, pcs))
    return out


def run_steps(pred, batches, n, first=None):
    """n pipelined steps over the rotating batches; returns the outputs and checks each against the first output seen for
    the same batch index (`first`, filled on the way)."""
    import numpy as np
    first = {} if first is None else first
    pending = []

    def take(j, ticket):
        s, m = ticket.wait()
        assert np.isfinite(s).all() and np.isfinite(m).all()
        if j not in first:
            first[j] = (s, m)
        else:
            assert np.array_equal(s, first[j][0]) and np.array_equal(m, first[j][1]), f"batch {j}: steps disagree"

    for i in range(n):
        if len(pending) >= 2:  # pinned output ring of 3: the slot reused next must have been consumed
            take(*pending.pop(0))
        j = i % len(batches)
        pending.append((j, pred.submit(*batches[j])))
    for p in pending:
        take(*p)
    return first


def isolated_xyz_search_ms(pred, iters=12):
    """The xyz-library distance GEMM of the LAST step once more, alone on an idle chip (same operands: the step's compacted query
    rows and live count, the same library operand, the same launch), HIP events around each launch: the kernel's own duration.
    Inside the pipelined step the searches run on the second stream beside the next step's extraction, where the measured
    duration also contains the time the kernel spends sharing the CUs (`roofline.launch_ms_in_pipeline`)."""
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    from cmdiad_amd.predictor import EventTimer
    torch.cuda.synchronize()
    bank = pred.bank_xyz
    t = EventTimer()
    ss = pred.static.get("ss_xyz_0")
    qs = pred.sets[0]["qs"] if pred.sets else None
    for _ in range(iters):
        if ss is not None:                       # row-sharded: the segments launch over the gathered live rows of all ranks
            ss.gemm(t)
        elif qs is not None and qs.get("xyz_plan") is not None:
            plan = qs["xyz_plan"]
            kc = ops.new_keys(plan.q16.shape[0], plan.q16.device)
            with t:
                ops.l2_min_keys_counted(plan.q16, plan.q_sq, plan.count, bank.bf16, bank.sqnorm, kc, bank.row_offset)
        elif qs is not None:
            _, q16, qsq = qs["xyz"]
            k = ops.new_keys(q16.shape[0], q16.device)
            with t:
                ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, k, bank.row_offset)
        else:
            return None
        torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in t.pairs[2:])      # the first two launches follow the pipeline's last steps: skipped
    return v[len(v) // 2]                                        # median of ten: one launch beside a late D2H copy must not move it


# --------------------------------------------------------------------------------------------------------- secondary legs
def profiled_traffic():
    """roofline.traffic: fabric-side bytes per launch of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md).  PMC
    counters cannot be read inside this process; the figure comes from the newest committed rocprofv3 --pmc pass
    (profiles/rN_pmc.json, tools/profile_round.sh) and is emitted ONLY while the kernel's source is byte-identical to the one that
    was profiled (profiles/rN_pmc_meta.json holds the sha256 of csrc/l2min.hip + gemm_core.h at that time): null as soon as the
    kernel changes."""
    import glob
    import hashlib
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    metas = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_meta.json")),
                   key=lambda f: int(re.search(r"r(\d+)_pmc_meta", f).group(1)), reverse=True)
    note = "no committed PMC pass"
    for meta_path in metas:
        tag = re.search(r"(r\d+)_pmc_meta", meta_path).group(1)
        try:
            meta = json.load(open(meta_path))
            h = hashlib.sha256()
            for f in meta["sources"]:
                h.update(open(os.path.join(here, f), "rb").read())
            if h.hexdigest() != meta["sha256"]:
                note = f"the distance GEMM's source changed since profiles/{tag}_pmc.json was taken: re-profile"
                continue
            if meta.get("standalone"):      # the launch ALONE on the chip: the regime `frac` / `launch_ms` are quoted in
                row = next(r for r in json.load(open(os.path.join(here, meta["standalone"]))) if r["shape"] == "bench")
                regime = "stand-alone launch of the bench's shape (tools/standalone_kernels.py l2), as `frac` / `launch_ms`"
                where = meta["standalone"]
            else:                           # the pipelined bench run (overlapped and isolated launches averaged)
                rows = [r for r in json.load(open(os.path.join(here, "profiles", f"{tag}_pmc.json"))) if r["kernel"].startswith("l2_min_pp3")]
                row = max(rows, key=lambda r: r["grid_threads"])
                regime = "inside the pipelined bench run"
                where = f"profiles/{tag}_pmc.md"
            return {"traffic": round(row["fetch_bytes"] + row["write_bytes"]), "traffic_regime": regime,
                    "traffic_note": f"fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from the committed PMC passes ({where}, profiles/{tag}_pmc.md, "
                                    f"commit {meta['commit']}; kernel source unchanged since: sha256 {meta['sha256'][:12]}); L2 hit {row['l2_hit']:.3f}"}
        except (OSError, KeyError, ValueError) as e:
            note = f"no usable committed PMC pass ({type(e).__name__})"
    return {"traffic": None, "traffic_note": note}


def cpu_baseline(n_images=10, warm=3):
    """The CPU oracle pipeline (oracle/pipeline.py, kind 'port': the reference's own torch-CPU composition + the C restatement
    of FPS / kNN) on a bounded sample of the same workload, on this box's host cores: at the thread count that is fastest
    here (`value`), at the reference's default --cpu_core_num 6 (main.py:149) and at all cores."""
    import torch
    from cmdiad_amd.models.models import PointTransformer, VisionTransformer
    from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb
    from oracle.pipeline import CpuDoubleRGBPoint, CpuExtractor
    torch.manual_seed(0)
    sd_vit = {k: v.detach() for k, v in VisionTransformer().state_dict().items()}
    sd_pm = {k: v.detach() for k, v in PointTransformer().state_dict().items()}
    cpu = CpuDoubleRGBPoint(CpuExtractor(sd_vit, sd_pm))
    cpu.set_banks(synth_bank(XYZ_ROWS, 768, 4321), synth_bank(RGB_ROWS, 768, 4322), 0.0, 1.0, 0.0, 1.0)
    all_threads = torch.get_num_threads()

    def timed(threads, n, w):
        torch.set_num_threads(threads)
        for i in range(w):
            cpu.predict(synth_rgb(i), synth_cloud_fixed_n(1000 + i, N_POINTS))
        cpu.ex.timing.clear(); cpu.timing.clear()
        t0 = time.perf_counter()
        for i in range(n):
            cpu.predict(synth_rgb(w + i), synth_cloud_fixed_n(1000 + w + i, N_POINTS))
        dt = time.perf_counter() - t0
        stages = {k: round(v / n, 4) for k, v in {**cpu.ex.timing, **cpu.timing}.items()}
        return n / dt, stages

    # torch's intra-op pool oversubscribes badly beyond ~32 threads on these shapes (128-thread MI355X host, round 1: 0.18
    # images/s at 128 threads, 0.47 at 32, 0.35 at 6), so the best setting is measured, not assumed
    best_t = min(32, all_threads)
    v_best, stages = timed(best_t, n_images, warm)
    v_six, _ = timed(min(6, all_threads), max(3, n_images // 3), 1)
    v_all, _ = (v_best, None) if all_threads == best_t else timed(all_threads, max(3, n_images // 3), 1)
    torch.set_num_threads(all_threads)
    return dict(value=round(v_best, 4), unit="images/s", cores=best_t, kind="port",
                sample=f"{n_images} images after {warm} warm-up, B=1, fp32, torch {torch.__version__} CPU ({best_t} intra-op threads: "
                       f"the fastest setting on this {all_threads}-thread host) + C oracle for FPS/kNN, same synthetic inputs and "
                       f"bagel-sized banks",
                at_reference_default_6_threads=round(v_six, 4), at_all_threads={"threads": all_threads, "value": round(v_all, 4)},
                seconds_per_image_by_stage=stages)


def dropin_b1(n=64, warm=16):
    """images/s of the B = 1 drop-in protocol (what the reference's main.py / cmdiad_runner.py drive):
    DoubleRGBPointFeatures.predict per image, host-resident samples (H2D of the sample and D2H of the maps included),
    bagel-sized libraries."""
    import types
    import warnings
    import numpy as np
    import torch
    from sklearn import linear_model
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures
    from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb
    a = dict(rgb_backbone_name='vit_base_patch8_224_dino', xyz_backbone_name='Point_MAE', group_size=128, num_group=1024,
             rgb_size=224, xyz_size=224, gt_size=224, f_coreset=1.0, coreset_eps=0.9, coreset_dtype='FP16',
             random_state=None, dist_method_s='l2', dist_method_coreset='l2', main_modality='', use_hn=False,
             fusion_module_path='', ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0, xyz_smap_lambda=1.0,
             rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
             save_feature_for_fusion=False, save_seg_results=False, use_depth=False)
    threads = torch.get_num_threads()
    torch.set_num_threads(6)  # main.py:149,190-191: the reference's default --cpu_core_num
    os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")  # synthetic weights: no checkpoints offline
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DoubleRGBPointFeatures(types.SimpleNamespace(**a))
    m.patch_xyz_lib = synth_bank(XYZ_ROWS, 768, 4321).cuda()
    m.patch_rgb_lib = synth_bank(RGB_ROWS, 768, 4322).cuda()
    m.xyz_mean = m.rgb_mean = torch.tensor(0.0)
    m.xyz_std = m.rgb_std = torch.tensor(1.0)
    rs = np.random.RandomState(0)
    m.detect_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 2))
    m.seg_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 2))
    samples = [(synth_rgb(i), synth_cloud_fixed_n(1000 + i, N_POINTS)) for i in range(8)]
    mask = torch.zeros(1, 224, 224)
    for i in range(warm):
        rgb, pc = samples[i % 8]
        m.predict((rgb, pc, pc), mask, 0, ["x.png"])
    assert len(m.image_preds) == warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        rgb, pc = samples[i % 8]
        m.predict((rgb, pc, pc), mask, 0, ["x.png"])
    assert len(m.image_preds) == warm + n      # reading a result attribute runs the last (partial) micro-batch
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    torch.set_num_threads(threads)
    return dict(value=round(n / dt, 2), unit="images/s", ms_per_image=round(dt / n * 1e3, 3),
                what=f"DoubleRGBPointFeatures.predict called once per image as cmdiad_runner.py drives it, {n} images after {warm} "
                     f"warm-up, host-resident samples, bagel-sized libraries, 6 host threads; the drop-in defers the calls "
                     f"into micro-batches of CMDIAD_PREDICT_BATCH={os.environ.get('CMDIAD_PREDICT_BATCH', '16')} (1 = strictly per call)")


def sharded_search(dev, group, rank, world, rows_list, iters=10, warm=3):
    """configs[3]: the patch-library nearest-neighbour search with the library's ROWS sharded over the ranks
    (cmdiad_amd.engine.ShardedSearch).  Every rank brings the 16-bit queries of its own batch of 32 images (100 352 x 768, 45.8 %
    of the rows the repeated background row, as in the bench's clouds); one iteration = local de-duplication -> counts exchange
    -> all-gather of the LIVE rows only -> distance GEMM of all ranks' live rows against this rank's row shard -> ONE
    integer-MIN all-reduce of the packed keys (RCCL over xGMI) -> expansion to one key per original row.  Iteration i + 1's
    exchange is issued on a second stream under iteration i's GEMM.  Timed with a barrier on both sides, max over ranks;
    the serial split (gather / GEMM / reduce + expand, HIP events, un-overlapped) is measured in a separate pass."""
    import types
    import torch
    import torch.distributed as td
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    Q = BATCH * 3136
    g = torch.Generator(device=dev).manual_seed(977 + rank)
    q32 = torch.randn(Q, 768, generator=g, device=dev)
    bg = torch.rand(Q, generator=g, device=dev) < (1.0 - N_POINTS / 50176.0) * 0.9   # patches without a foreground pixel
    q32[bg] = -0.3
    q16, _, qsq = ops.normalize_cast(q32)
    del q32
    side = ops.shared_stream(dev, "bench.exchange")
    out = []
    for name, rows in rows_list:
        lo, hi = eng.shard_range(rows, rank, world)
        gb = torch.Generator(device=dev).manual_seed(4321 + rows)  # every rank draws the same library, keeps its rows
        full = torch.randn(rows, 768, generator=gb, device=dev)
        b16, _, bsq = ops.normalize_cast(full[lo:hi].contiguous())
        del full
        bank = types.SimpleNamespace(bf16=b16, sqnorm=bsq, row_offset=lo)
        stats = {}
        searches = [eng.ShardedSearch(bank, group, stats=stats) for _ in range(2)]
        cur = torch.cuda.current_stream()

        def gather_on_side(s, after):
            side.wait_event(after)        # NOT wait_stream(cur): the GEMM just queued on `cur` is what this exchange runs under
            with torch.cuda.stream(side):
                s.gather(q16, qsq)

        def mark():
            e = torch.cuda.Event()
            e.record(cur)
            return e

        def run(n):
            gather_on_side(searches[0], mark())
            keys = None
            for i in range(n):
                s = searches[i & 1]
                cur.wait_stream(side)                 # this iteration's exchange has landed
                before_gemm = mark()                  # everything up to the previous iteration's reduce: the other buffer set is free
                s.gemm()
                for t in (s.q_all, s.s_all):          # allocated on `side`, read on `cur`
                    t.record_stream(cur)
                if i + 1 < n:
                    gather_on_side(searches[(i + 1) & 1], before_gemm)   # the next exchange, under this GEMM; its collectives are
                keys = s.reduce()                                        # queued before this iteration's min-reduce
            return keys

        merged = run(warm)
        assert int((merged == eng.KEY_EMPTY).sum()) == 0           # every query found a row somewhere
        td.barrier(group)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(iters)
        torch.cuda.synchronize()
        td.barrier(group)
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        td.all_reduce(dt, op=td.ReduceOp.MAX, group=group)
        ms = float(dt.item()) / iters * 1e3
        # the serial split: the three stages one after the other on one stream, HIP events between them
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(3)]
        for e4 in ev:
            s = searches[0]
            e4[0].record(); s.gather(q16, qsq); e4[1].record(); s.gemm(); e4[2].record(); s.reduce(); e4[3].record()
        torch.cuda.synchronize()
        split = [sum(e4[k].elapsed_time(e4[k + 1]) for e4 in ev) / len(ev) for k in range(3)]
        live = sum(stats["live_rows"])
        flops = 2.0 * live * (hi - lo) * 768
        out.append(dict(cls=name, rows=rows, rows_this_rank=hi - lo, ms_per_search=round(ms, 3),
                        images_per_s=round(world * BATCH / (ms * 1e-3), 1),
                        serial_ms_rank0=dict(dedup_and_gather=round(split[0], 3), gemm=round(split[1], 3), reduce_and_expand=round(split[2], 3)),
                        gemm_tflops_rank0=round(flops / (split[1] * 1e-3) / 1e12, 1),
                        overlap_gain_ms=round(sum(split) - ms, 3),
                        live_rows_per_rank=stats["live_rows"], gathered_rows_per_rank=stats["gathered_rows_per_rank"],
                        gather_MB_received_per_rank=round(stats["gather_bytes_received"] / 1e6, 2),
                        gather_MB_received_without_compaction=round(stats["gather_bytes_received_without_compaction"] / 1e6, 2),
                        reduce_MB=round(stats["reduce_bytes"] / 1e6, 3)))
        del b16, bsq, searches
    return dict(what="row-sharded library search: local de-duplication of the repeated background row -> all-gather of the live 16-bit "
                     "query rows only -> per-shard distance GEMM -> one all_reduce(MIN) of packed int64 keys -> expansion; the next "
                     "iteration's exchange runs under the current GEMM; weak scaling, 32 images (100 352 query rows) per rank",
                rccl_ranks=td.get_world_size(group), backend=td.get_backend(group), classes=out)


def fake_world_leg(dev, classes=("bagel", "peach"), worlds=(1, 2, 4, 8), iters=4):
    """configs[3] at its REAL shard shapes, on one GPU ("fake world"): for W in `worlds` the library's rows are cut
    into the W shards `engine.Bank` makes (128-row aligned, search operand padded to whole tiles), W separately compacted query sets
    of 32 images each (100 352 rows, 45.8 % of them the repeated background row) are laid out as the gathered operand of
    `engine.ShardedSearch` (W segments of `cap` rows + the W live counts on the device), and EVERY shard's distance GEMM -- one
    `cmdiad_l2_min_keys_segments` launch, what one rank of a W-rank node executes per step -- is timed alone with HIP events.  The
    integer MIN over the W shards' keys is compared with the single-library keys (bit for bit).  The exchange is NOT measured here
    (one GPU): gather / reduce bytes are stated and a link model turns them into a predicted per-search time and rate."""
    import torch
    from cmdiad_amd import engine as eng
    from cmdiad_amd import ops
    Q, D = BATCH * 3136, 768
    LINK_GBS, LINK_EFF, COLL_LAT_US = 153.0, 0.8, 30.0      # xGMI: one link per peer, 153 GB/s per direction (MI355X_MICROARCH.md)
    g = torch.Generator(device=dev).manual_seed(977)
    q32 = torch.randn(Q, D, generator=g, device=dev)
    bg = torch.rand(Q, generator=g, device=dev) < (1.0 - N_POINTS / 50176.0) * 0.9   # patches without a foreground pixel
    q32[bg] = -0.3
    q16, _, qsq = ops.normalize_cast(q32)
    del q32
    wmax = max(worlds)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plans = []
    for w in range(wmax):                    # rank w's own batch: the same rows in another order (its own compaction)
        sh = (w * 9973) % Q
        plans.append(ops.rows_dedup_plan(torch.roll(q16, sh, 0).contiguous(), torch.roll(qsq, sh, 0).contiguous()))
    e1.record()
    torch.cuda.synchronize()
    dedup_ms = e0.elapsed_time(e1) / wmax    # incl. the roll; an upper bound of the plan's ~0.12 ms
    counts = [int(p.count.item()) for p in plans]
    cap = min(Q, (max(counts) + 255) // 256 * 256)
    row_bytes = D * 2 + 4
    out = []
    for name in classes:
        rows = class_rows(name)
        gb = torch.Generator(device=dev).manual_seed(4321 + rows)
        full = torch.randn(rows, D, generator=gb, device=dev)
        whole = eng.Bank(full, 0, 1)             # the single-library answer for rank 0's live rows (the counted launch of the pipeline)
        ref_keys = ops.l2_min_keys_counted(plans[0].q16, plans[0].q_sq, plans[0].count, whole.bf16, whole.sqnorm, ops.new_keys(Q, dev))
        del whole
        for W in worlds:
            q_all = torch.cat([p.q16[:cap] for p in plans[:W]])
            s_all = torch.cat([p.q_sq[:cap] for p in plans[:W]])
            cnt = torch.tensor(counts[:W], dtype=torch.int32, device=dev)
            merged = None
            ms = []
            for r in range(W):
                bank = eng.Bank(full, r, W)
                keys = ops.new_keys(W * cap, dev)
                ops.l2_min_keys_segments(q_all, s_all, cnt, cap, bank.bf16, bank.sqnorm, keys, bank.row_offset)   # warm + the checked result
                merged = keys if merged is None else torch.minimum(merged, keys)
                scratch = ops.new_keys(W * cap, dev)
                t = 0.0
                for _ in range(iters):
                    scratch.fill_(eng.KEY_EMPTY)
                    e0.record()
                    ops.l2_min_keys_segments(q_all, s_all, cnt, cap, bank.bf16, bank.sqnorm, scratch, bank.row_offset)
                    e1.record()
                    torch.cuda.synchronize()
                    t += e0.elapsed_time(e1)
                ms.append(t / iters)
                shard_rows, shard_tiles = bank.shard_rows, bank.bf16.shape[0] // 256
                del bank, keys, scratch
            same = bool(torch.equal(merged[:counts[0]], ref_keys[:counts[0]]))
            live = sum(counts[:W])
            per = ((rows + W - 1) // W + 127) // 128 * 128
            flops = 2.0 * live * min(per, rows) * D      # the largest (= every but the last) shard
            gemm = max(ms)
            gather_b = (W - 1) * cap * row_bytes
            reduce_b = W * cap * 8
            t_gather = cap * row_bytes / (LINK_GBS * 1e9 * LINK_EFF) * 1e3 + COLL_LAT_US * 1e-3 if W > 1 else 0.0   # every peer's segment over its own link
            t_reduce = (2.0 * (W - 1) / W * reduce_b / (min(W - 1, 7) * LINK_GBS * 1e9 * LINK_EFF) * 1e3 + COLL_LAT_US * 1e-3) if W > 1 else 0.0
            t_search = dedup_ms + max(gemm, t_gather) + t_reduce
            out.append(dict(cls=name, rows=rows, world=W, rows_per_rank=min(per, rows), shard_tiles=shard_tiles, live_rows_per_rank=counts[:W],
                            gathered_rows_per_rank=cap, gemm_ms_slowest_rank=round(gemm, 3), gemm_ms_mean=round(sum(ms) / len(ms), 3),
                            gemm_tflops_per_rank=round(flops / (gemm * 1e-3) / 1e12, 1), gemm_frac_of_peak=round(flops / (gemm * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                            merged_keys_equal_single_library=same,
                            gather_MB_received_per_rank=round(gather_b / 1e6, 2), reduce_MB=round(reduce_b / 1e6, 3),
                            model=dict(gather_ms=round(t_gather, 3), reduce_ms=round(t_reduce, 3), search_ms=round(t_search, 3),
                                       images_per_s=round(W * BATCH / (t_search * 1e-3), 1))))
            assert same, f"fake world {name} W={W}: the MIN over the shards' keys differs from the single-library keys"
        del full
    return dict(what="compute side measured on 1 GPU, links not measured: per W the distance GEMM of ONE rank of a W-rank node (all W ranks' live "
                     "query rows against a 1/W row shard, one cmdiad_l2_min_keys_segments launch, HIP events, every shard timed in turn); "
                     "model = dedup + max(GEMM, all-gather) + all-reduce with one xGMI link per peer",
                link_model=dict(link_GBs_per_direction=LINK_GBS, efficiency=LINK_EFF, collective_latency_us=COLL_LAT_US),
                dedup_ms=round(dedup_ms, 3), shapes=out)


def train_step_leg(dev, steps=50, warm=10):
    """BASELINE configs[2]: one FtoF distillation training step = both directions forward + loss + backward + Adam on a
    [32, 3136, 1536] feature batch (xyz first, rgb second), N(0,1), seed 3407 (hallucination_network_pretrain.py:53,102-159), lr
    schedule per iteration (utils/lr_sched.py:4-17), l2 loss; 7.99 TFLOP per step (3 x forward, both directions, 100 352
    tokens).  The batch is resident in HBM (tools/train_bench.py also times the FeatureRing-fed loop)."""
    import types
    import torch
    from cmdiad_amd import train
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork
    from cmdiad_amd.utils import lr_sched
    torch.manual_seed(3407)
    net = HallucinationCrossModalityNetwork(None, 768, 768).to(dev)
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    sched = types.SimpleNamespace(lr=5e-4, warmup_epochs=10, epochs=100)
    x = torch.randn(32, 3136, 1536, generator=torch.Generator(device=dev).manual_seed(3407), device=dev)
    losses = []

    def step(it):
        lr_sched.adjust_learning_rate(opt, it / 100.0, sched)
        lx, lr_ = net(x[:, :, :768], x[:, :, 768:], False, "l2")
        opt.zero_grad(set_to_none=True)
        (lx + lr_).backward()
        opt.step()
        return lx, lr_

    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        lx, lr_ = step(warm + i)
        if i in (0, steps - 1):
            losses.append((lx, lr_))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    l0, l1 = [float(a.item() + b.item()) for a, b in losses]
    assert l1 == l1 and l1 < l0, (l0, l1)            # finite, and the optimiser is descending
    return dict(what="configs[2]: HallucinationCrossModality feature-to-feature distillation training step (forward + l2 loss + "
                     "backward, both directions, + Adam) on [32, 3136, 1536] synthetic features resident in HBM",
                ms_per_step=round(dt * 1e3, 3), steps_per_s=round(1.0 / dt, 2), tflop_per_step=7.99,
                achieved_TFLOPs=round(7.99 / dt, 1), frac_of_mfma_peak=round(7.99 / dt / PEAK_BF16_TFLOPS, 4),
                steps=steps, warmup=warm, loss_first_timed=round(l0, 2), loss_last_timed=round(l1, 2),
                tokens_per_s=round(32 * 3136 / dt, 0))


def conv_head_train_leg(dev, batch=8, steps=6, warm=2):
    """one training step of the convolutional FtoF head (HallucinationCrossModalityConv: per direction conv3x3 ->
    batch-statistics BatchNorm -> ReLU three times + conv3x3, hallucination_network.py:72-147) -- both directions, forward + l2 loss +
    backward + Adam -- on the hand-written path of cmdiad_amd/conv_train.py.  FLOPs: 2 towers x (4 forward + 3 data-gradient + 4
    weight-gradient convolutions) x 2 M 768 (9 768), M = batch x 3136 positions."""
    import torch
    from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityConv
    torch.manual_seed(3407)
    net = HallucinationCrossModalityConv(None, 768, 768).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(3407)
    a, b = torch.randn(batch, 3136, 768, generator=g, device=dev), torch.randn(batch, 3136, 768, generator=g, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        lx, lr_ = net(a, b, False, "l2")
        (lx + lr_).backward()
        opt.step()
        return lx, lr_

    first = None
    for i in range(warm):
        lx, lr_ = step()
        first = first if first is not None else float(lx.detach() + lr_.detach())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lx, lr_ = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    last = float(lx.detach() + lr_.detach())
    assert last == last and last < first, (first, last)
    tflop = 2 * 11 * 2.0 * batch * 3136 * 768 * 9 * 768 / 1e12
    return dict(what="row f4: HallucinationCrossModalityConv training step (both directions: forward, l2 loss, backward, Adam), "
                     "hand-written HIP forward + backward (cmdiad_amd/conv_train.py), batch-statistics BatchNorm",
                batch=batch, ms_per_step=round(dt * 1e3, 2), tflop_per_step=round(tflop, 2), achieved_TFLOPs=round(tflop / dt, 1),
                steps=steps, warmup=warm, loss_first=round(first, 1), loss_last=round(last, 1))


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
                         **profiled_traffic(),
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

    def leg(name, fn, seconds, collective=False):
        return legs.run(name, fn, budget or seconds, collective=collective)

    if not args.no_extras:
        n_h2d = max(8, min(args.steps, 12))

        def h2d_leg():
            # PCIe-inclusive rate: the same batches from pinned host memory, H2D on the predictor's copy stream inside the loop
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(pred, host_batches, n_h2d, first)   # also checks H2D-fed outputs == resident-fed outputs
            torch.cuda.synchronize()
            h2d_dt = time.perf_counter() - t1
            mb = sum(t.numel() * 4 for t in host_batches[0] if t is not None) / 1e6
            return dict(value=round(BATCH * n_h2d / h2d_dt, 2), unit="images/s per GPU", steps=n_h2d, h2d_MB_per_step=round(mb, 1),
                        note="inputs in pinned host memory, copied inside the loop; outputs identical to the resident run")

        def every_row_leg():
            # the same steps with EVERY row of the patch grid searched, as the reference's cdist does (no row de-duplication):
            # outputs are checked bit for bit against the de-duplicated run's (run_steps compares with `first`)
            os.environ["CMDIAD_DEDUP"] = "0"
            try:
                pred_all = BatchPredictor(st["engine"], st["bank_xyz"], st["bank_second"], st["stats"], st["det"], st["seg"], batch=BATCH,
                                          n_max=N_POINTS, workload=args.workload, halluc=st["halluc"], group=None,
                                          use_graph=pred.use_graph)
            finally:
                del os.environ["CMDIAD_DEDUP"]
            run_steps(pred_all, batches, 3, first)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            run_steps(pred_all, batches, n_h2d, first)
            torch.cuda.synchronize()
            all_dt = time.perf_counter() - t2
            return dict(value=round(BATCH * n_h2d / all_dt, 2), unit="images/s per GPU", steps=n_h2d,
                        ms_per_step=round(all_dt / n_h2d * 1e3, 3),
                        note="CMDIAD_DEDUP=0: all 100352 query rows per step go through the distance GEMM; "
                             "outputs bit-identical to the default run")

        # (with the row-sharded pipeline -- --bank sharded -- the H2D-fed steps are collective steps too)
        leg("h2d_inclusive", h2d_leg, 120, collective=sharded)
        if group is None and pred.dedup:
            leg("every_row_searched", every_row_leg, 180)
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
        sys.stdout.flush()
        os._exit(0)      # some rank failed a leg: the others may sit in a collective that never completes -- no orderly teardown


if __name__ == "__main__":
    main()
