"""Class-sharded evaluation: the loop of the reference's ``main.py:22-37`` (for every class a fresh ``CMDIAD`` object,
``fit`` then ``evaluate``, cmdiad_runner.py:33-107) with the classes dealt to the GPUs of one node.

BASELINE configs[4] / SURVEY 8(e) "class sharding": classes are independent of each other (a fresh method object and
``set_seeds(0)`` per class, main.py:23, features.py:48), so a rank evaluates its classes start to finish -- memory bank,
coreset, late-fusion bank, the two one-class SVMs, predict, metrics -- with NO collective on the data path; the four metric
dictionaries (image ROCAUC, pixel ROCAUC, AU-PRO, AU-PRO 0.01; cmdiad_runner.py:87-107) are gathered to every rank at the
end and rank 0 builds the table main.py prints (per-class columns + ``Mean`` rounded to three digits, main.py:34-37).

The assignment is static longest-processing-time-first over a cost model of a class (greedy coreset rounds dominate a fit:
quadratic in the number of train images; extraction and scoring are linear): deterministic, the same on every rank, no
communication needed to agree on it.

Everything below the protocol calls is the drop-in classes of ``feature_extractors/multiple_features.py`` (HIP kernels);
this module holds no arithmetic of its own.
"""
import os
import time
import types

import numpy as np
import torch

# MVTec 3D-AD split sizes [external counts, SURVEY 8d]: 2 656 train / 1 197 test images in ten classes
MVTEC3D_TRAIN = {"bagel": 244, "cable_gland": 223, "carrot": 286, "cookie": 210, "dowel": 288, "foam": 236, "peach": 361,
                 "potato": 300, "rope": 298, "tire": 210}
MVTEC3D_TEST = {"bagel": 110, "cable_gland": 108, "carrot": 159, "cookie": 131, "dowel": 130, "foam": 100, "peach": 132,
                "potato": 114, "rope": 101, "tire": 112}
METRICS = ("image_rocauc", "pixel_rocauc", "au_pro", "au_pro_001")


def mtfi_args(**kw):
    """The ``args`` namespace of main.py:85-189 with the MTFI feature-to-feature settings of configs[4]
    (``--method_name WithHallucination --use_hn --main_modality xyz``; README.md of the reference)."""
    a = dict(method_name="WithHallucination", rgb_backbone_name="vit_base_patch8_224_dino", xyz_backbone_name="Point_MAE",
             group_size=128, num_group=1024, rgb_size=224, xyz_size=224, gt_size=224, f_coreset=0.1, coreset_eps=0.9,
             coreset_dtype="FP16", random_state=None, dist_method_s="l2", dist_method_coreset="l2", main_modality="xyz",
             use_hn=True, fusion_module_path="", ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0, xyz_smap_lambda=1.0,
             rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0, memory_bank="multiple",
             max_sample=500, train_with_validation=False, save_feature_for_fusion=False, save_seg_results=False,
             use_depth=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def method_class(args):
    """cmdiad_runner.py:16-31: method name -> (key of the metric dictionaries, drop-in class)."""
    from .feature_extractors import multiple_features as mf
    table = {"DINO": mf.RGBFeatures, "Point_MAE": mf.PointFeatures, "DINO+Point_MAE": mf.DoubleRGBPointFeatures,
             "WithHallucination": mf.RGBorXYZWithOneHallucination,
             "WithHallucinationFromFeature": mf.RGBorXYZWithOneHallucinationFromFeature}
    if args.method_name not in table:
        raise ValueError(f"unknown method_name {args.method_name!r} (cmdiad_runner.py:16-31 knows {sorted(table)})")
    return args.method_name, table[args.method_name]


# ------------------------------------------------------------------------------------------------ assignment
def class_cost(n_train, n_test, f_coreset=0.1, libraries=2, rows_per_image=3136, t_image=2.0e-3, t_row_round=1.18e-10,
               t_proj_row=1.0e-6, t_svm_row=1.0e-7):
    """Estimated seconds one GPU (and its host thread) spends on a class: two passes over the train images (memory bank,
    late-fusion bank) and one over the test images at ``t_image`` each; per library the sparse random projection on the host
    (``t_proj_row`` per library row) and a greedy coreset of f*rows rounds each scanning all rows (csrc/coreset.hip: 90 us per
    round at 765 184 rows, profiles/r2_notes.md -> 1.18e-10 s per row and round); the host fit of the pixel-level one-class SVM over
    n_train * 224^2 rows, ~9 epochs (docs/history.md 7: 11.4 s at 12.2 M rows).  Calibrated on `bench.py --evaluate` (profiles/r3_notes.md:
    61 / 90 train images -> 4.7 / 8.4 s bank + coreset, 3.4 / 5.8 s late fusion); only the ORDER of the costs matters to the
    assignment."""
    rows = n_train * rows_per_image
    coreset = libraries * ((f_coreset * rows) * rows * t_row_round + rows * t_proj_row) if f_coreset < 1 else 0.0
    return (2 * n_train + n_test) * t_image + coreset + 9 * n_train * 50176 * t_svm_row


def lpt_assign(costs, world):
    """Longest-processing-time-first: classes by decreasing cost (ties: name), each to the rank with the least load so far
    (ties: lowest rank).  -> list of ``world`` lists of class names; every class appears exactly once.  Deterministic, so
    every rank computes the same assignment without talking to the others."""
    if world < 1:
        raise ValueError("world must be >= 1")
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for name in sorted(costs, key=lambda n: (-float(costs[n]), n)):
        r = min(range(world), key=lambda i: (loads[i], i))
        out[r].append(name)
        loads[r] += float(costs[name])
    return out


# ------------------------------------------------------------------------------------------------ one class
class ClassRun:
    """fit + evaluate of ONE class, in the reference's order (cmdiad_runner.py:33-107), cut into the three stages a rank can
    overlap ACROSS classes (the order inside a class never changes):

      fit_device():  add_sample_to_mem_bank over the train loader -> run_coreset -> (memory_bank == 'multiple')
                     add_sample_to_late_fusion_mem_bank over the train loader again, scored on the GPU   [device-bound]
      fit_host():    run_late_fusion: the two scikit-learn SGDOneClassSVM fits of features.py:352-358 -- 8-9 s of ONE host core
                     per full-size class, during which the GPU has nothing to do for THIS class             [host-bound]
      predict():     predict over the test loader -> calculate_metrics.

    ``data``: an object with ``name``, ``train()`` yielding (sample, label) and ``test()`` yielding (sample, mask, label,
    rgb_path) -- the reference's loaders, or synth.SyntheticClass.  ``weights`` (optional): (ViT state_dict, Point-MAE
    state_dict, fusion state_dict | None) loaded into the fresh method object (offline stand-in for the checkpoints).
    ``extractor`` (optional): the frozen backbones (``Features.deep_feature_extractor``) of an earlier class on this rank --
    every other piece of state (banks, statistics, SVMs, result lists) is the fresh object's."""

    def __init__(self, args, data, weights=None, method=None, extractor=None):
        from .utils.utils import set_seeds
        if method is None:
            _, cls = method_class(args)
            method = cls(args, shared_extractor=extractor)      # a fresh object per class, as main.py:23
        if weights is not None and extractor is None:
            method.deep_feature_extractor.rgb_backbone.load_state_dict(weights[0])
            method.deep_feature_extractor.xyz_backbone.load_state_dict(weights[1])
        if weights is not None and len(weights) > 2 and weights[2] is not None and getattr(method, "fusion", None) is not None:
            method.fusion.load_state_dict(weights[2])
        set_seeds(0)
        self.args, self.data, self.method = args, data, method
        self.count = getattr(args, "max_sample", 500)
        self.sec, self.phases = {}, []
        self.multiple = getattr(args, "memory_bank", "multiple") == "multiple"
        self.n_train = self.n_test = 0

    def _loop(self, it, call, limit):
        n = flag = 0
        for item in it:
            call(item)
            n += 1
            flag += 1
            if limit is not None and flag > limit:           # cmdiad_runner.py:50-52, 64-66: the two TRAIN loops stop AFTER max_sample + 1 samples
                break
        return n

    def fit_device(self):
        m, data = self.method, self.data
        # the loaders are drained first (the reference's DataLoader workers prefetch beside the model; a synthetic class generates its
        # samples on this thread): "load" is reported on its own and the phase timings below are the method's
        t0 = time.perf_counter()
        self.train_items = list(data.train())
        self.test_items = list(data.test())
        self.sec["load"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.n_train = self._loop(self.train_items, lambda it: m.add_sample_to_mem_bank(it[0], class_name=data.name), self.count)
        m.run_coreset()
        torch.cuda.synchronize()
        self.sec["memory_bank_and_coreset"] = time.perf_counter() - t0
        self.phases += ["memory_bank", "coreset"]
        if self.multiple:
            t0 = time.perf_counter()
            self._loop(self.train_items, lambda it: m.add_sample_to_late_fusion_mem_bank(it[0]), self.count)
            flush = getattr(m, "_flush", None)               # the drop-in defers the samples into micro-batches: score them now (GPU)
            if flush is not None:
                flush("late")
            torch.cuda.synchronize()
            self.sec["late_fusion_bank"] = time.perf_counter() - t0
            self.phases.append("late_fusion_bank")
        self.train_items = None
        return self

    def host_fit_is_host_only(self):
        """True when fit_host() touches no GPU state (scikit-learn fits): it may then run on a worker thread."""
        from sklearn.base import BaseEstimator
        m = self.method
        return self.multiple and all(isinstance(getattr(m, f, None), BaseEstimator) for f in ("detect_fuser", "seg_fuser"))

    def fit_host(self):
        if self.multiple:
            t0 = time.perf_counter()
            self.method.run_late_fusion()
            self.sec["late_fusion_fit"] = time.perf_counter() - t0
            self.sec["late_fusion"] = self.sec["late_fusion_bank"] + self.sec["late_fusion_fit"]
            self.phases.append("late_fusion_fit")
        return self

    def predict(self):
        m = self.method
        with torch.no_grad():
            t0 = time.perf_counter()
            self.n_test = self._loop(self.test_items, lambda it: m.predict(*it), None)   # cmdiad_runner.py:80-85: every test sample, no cut-off
            # the drop-in defers predict() into micro-batches; reading a result attribute completes them (multiple_features._MethodBase)
            assert len(m.image_preds) == self.n_test
            torch.cuda.synchronize()
            self.sec["predict"] = time.perf_counter() - t0
        self.phases.append("predict")
        t0 = time.perf_counter()
        m.calculate_metrics()
        self.sec["metrics"] = time.perf_counter() - t0
        self.phases.append("metrics")
        self.test_items = None
        return self

    def result(self):
        m = self.method
        out = {k: float(getattr(m, k)) for k in METRICS}
        out.update(n_train=self.n_train, n_test=self.n_test, seconds={k: round(v, 3) for k, v in self.sec.items()},
                   phases=list(self.phases),
                   library_rows={k: int(getattr(m, f"patch_{k}_lib").shape[0]) for k in ("xyz", "rgb", "fusion")
                                 if torch.is_tensor(getattr(m, f"patch_{k}_lib", None))})
        out["_extractor"] = m.deep_feature_extractor
        return out


def _limit_host_threads():
    """torch's CPU operators (staging the micro-batches, a synthetic class generating its samples: operators on ~150 k elements)
    run on an OpenMP pool of one thread per core whose idle threads SPIN between operators.  On a 128-thread host that pool makes
    those small operators 20x slower than 8 threads do (a synthetic class of 61 + 20 images: 2.5 s -> 0.13 s to generate) and
    slows the single-threaded SGD fit on the worker thread down by 2.5x (1.8 -> 4.3 s).  The host side of the class loop never
    needs more than a few threads: CMDIAD_EVAL_HOST_THREADS (default 8).  Returns the previous count."""
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(before, int(os.environ.get("CMDIAD_EVAL_HOST_THREADS", "8")))))
    return before


def run_class(args, data, weights=None, method=None, extractor=None):
    """The three stages of ClassRun in line.  Returns the class's metrics (unrounded), image counts, seconds per phase, the
    phase order, and the extractor under "_extractor"."""
    return ClassRun(args, data, weights, method, extractor).fit_device().fit_host().predict().result()


def run_classes_overlapped(args, datasets, names, weights=None, log=None):
    """The classes `names` of one rank with the host-bound stage of class k (the two one-class-SVM fits: one host core, no GPU)
    on a worker THREAD beside the device-bound stage of class k + 1 (memory bank, coreset, late-fusion bank):

        fit_device(c0) | fit_host(c0) on the worker || fit_device(c1) | join | predict(c0) | fit_host(c1) || fit_device(c2) | ...

    Inside a class the reference's order is untouched (cmdiad_runner.py:33-107); classes are independent of each other (a fresh
    method object per class, main.py:23; the worker uses no global RNG: SGDOneClassSVM(random_state=42), features.py:114-115), so
    every number equals the in-line run's.  No fork, no second GPU process: scikit-learn's SGD loop releases the GIL."""
    import threading
    out, extractor = {}, None
    pending = None            # (ClassRun, thread | None, error list)
    threads_before = _limit_host_threads()

    def finish(p):
        run, th, err = p
        if th is not None:
            th.join()
        if err:
            raise err[0]
        if th is None:
            run.fit_host()
        res = run.predict().result()
        res.pop("_extractor")
        out[run.data.name] = res
        if log is not None:
            log(f"class {run.data.name}: " + ", ".join(f"{m} {res[m]:.3f}" for m in METRICS) + f" {res['seconds']}")

    try:
        for cls in names:
            run = ClassRun(args, datasets[cls], weights=weights, extractor=extractor)
            extractor = run.method.deep_feature_extractor
            run.fit_device()
            if pending is not None:
                finish(pending)
            th, err = None, []
            if run.host_fit_is_host_only():
                def work(r=run, e=err):
                    try:
                        r.fit_host()
                    except BaseException as exc:      # re-raised on the main thread at the join
                        e.append(exc)
                th = threading.Thread(target=work, name=f"late-fusion-fit-{cls}", daemon=True)
                th.start()
            pending = (run, th, err)
        if pending is not None:
            finish(pending)
    finally:
        torch.set_num_threads(threads_before)
    return out


# ------------------------------------------------------------------------------------------------ all classes
def metrics_table(per_class, method_name):
    """main.py:27-37: one row per metric with a column per class (rounded to 3 digits, cmdiad_runner.py:98-101) and their
    ``Mean`` (rounded to 3 digits).  Columns follow the order of ``per_class``."""
    table = {}
    for m in METRICS:
        row = {cls.title(): round(per_class[cls][m], 3) for cls in per_class}
        row["Mean"] = round(float(np.mean(list(row.values()))), 3) if row else float("nan")
        table[m] = {"Method": method_name, **row}
    return table


def evaluate_classes(args, datasets, group=None, weights=None, costs=None, log=None, runner=None):
    """``datasets``: {class name: data object (see run_class)} -- the SAME dictionary on every rank.  With ``group`` (a
    torch.distributed process group: RCCL on the GPUs, gloo in the CPU tests) the classes are dealt to the ranks by
    ``lpt_assign`` over ``costs`` (default: ``class_cost`` of the class sizes), every rank runs its classes, and the
    per-class results are gathered to all ranks (``all_gather_object``: a few hundred bytes -- the only collective).
    ``runner`` (default ``run_class``) evaluates one class -- the CPU tests of the sharding logic pass a stand-in.
    Returns dict(per_class, table, assignment, rank_seconds, method)."""
    rank, world = 0, 1
    if group is not None:
        import torch.distributed as td
        rank, world = td.get_rank(group), td.get_world_size(group)
    if costs is None:
        costs = {n: class_cost(d.n_train, d.n_test, getattr(args, "f_coreset", 0.1)) for n, d in datasets.items()}
    assignment = lpt_assign(costs, world)
    name, _ = method_class(args)
    mine = {}
    extractor = None
    error = None
    t0 = time.perf_counter()
    overlap = runner is None and os.environ.get("CMDIAD_EVAL_OVERLAP", "1") != "0" and len(assignment[rank]) > 1
    threads_before = _limit_host_threads() if runner is None else None
    try:
        if overlap:   # the host SVM fits of class k beside the device work of class k + 1 (run_classes_overlapped)
            mine = run_classes_overlapped(args, datasets, assignment[rank], weights=weights,
                                          log=(lambda msg: log(f"[rank {rank}] {msg}")) if log is not None else None)
            for cls in mine:
                mine[cls]["rank"] = rank
        for cls in ([] if overlap else assignment[rank]):
            if runner is None:
                mine[cls] = run_class(args, datasets[cls], weights=weights, extractor=extractor)
                extractor = mine[cls].pop("_extractor")          # the frozen backbones stay on the rank; everything else is per class
            else:
                mine[cls] = runner(args, datasets[cls], weights=weights)
            mine[cls]["rank"] = rank
            if log is not None:
                log(f"[rank {rank}] class {cls}: " + ", ".join(f"{m} {mine[cls][m]:.3f}" for m in METRICS) + f" {mine[cls]['seconds']}")
    except Exception as exc:      # a rank that fails must still reach the gather: the others would wait for it until the collective times out
        if group is None:
            raise
        error = f"{type(exc).__name__}: {exc}"
    finally:
        if threads_before is not None:
            torch.set_num_threads(threads_before)
    mine_s = time.perf_counter() - t0
    parts = [(rank, mine, mine_s, error)]
    if group is not None:
        parts = [None] * world
        td.all_gather_object(parts, (rank, mine, mine_s, error), group=group)
    failed = [(r, e) for r, _, _, e in parts if e is not None]
    if failed:
        raise RuntimeError("class-sharded evaluation failed on " + "; ".join(f"rank {r}: {e}" for r, e in failed))
    merged, rank_seconds = {}, [0.0] * world
    for r, part, s, _ in parts:
        rank_seconds[r] = round(s, 3)
        for cls, res in part.items():
            if cls in merged:
                raise RuntimeError(f"class {cls} was evaluated by ranks {merged[cls]['rank']} and {r}")
            merged[cls] = res
    missing = [c for c in datasets if c not in merged]
    if missing:
        raise RuntimeError(f"classes not evaluated by any rank: {missing}")
    per_class = {c: merged[c] for c in datasets}          # the caller's class order, as main.py's loop
    return dict(method=name, per_class=per_class, table=metrics_table(per_class, name), assignment=assignment,
                estimated_cost_s={c: round(float(costs[c]), 3) for c in datasets}, rank_seconds=rank_seconds, world=world)


def synthetic_mvtec3d(classes="all", scale=1.0, n_train=None, n_test=None, severity=1.0):
    """{class: synth.SyntheticClass} with the MVTec 3D-AD split sizes times ``scale`` (or fixed ``n_train`` / ``n_test``);
    ``severity``: strength of the planted defects (synth.SyntheticClass)."""
    from .synth import SyntheticClass
    names = list(MVTEC3D_TRAIN) if classes == "all" else [c for c in classes if c]
    out = {}
    for c in names:
        if c not in MVTEC3D_TRAIN:
            raise ValueError(f"unknown MVTec 3D-AD class {c!r}")
        tr = n_train if n_train is not None else max(2, int(round(MVTEC3D_TRAIN[c] * scale)))
        te = n_test if n_test is not None else max(4, int(round(MVTEC3D_TEST[c] * scale)))
        out[c] = SyntheticClass(c, tr, te, index=list(MVTEC3D_TRAIN).index(c), severity=severity)
    return out
