"""Drop-in for the reference's ``feature_extractors/multiple_features.py``: the five method classes that
``cmdiad_runner.CMDIAD`` dispatches to (RGBFeatures :28-121, PointFeatures :207-309,
RGBorXYZWithOneHallucination :312-573, RGBorXYZWithOneHallucinationFromFeature :576-797,
DoubleRGBPointFeatures :800-1015), with the same five-call protocol:
add_sample_to_mem_bank / run_coreset / add_sample_to_late_fusion_mem_bank / run_late_fusion / predict.

The per-class differences of the reference (which banks exist, which scalar statistics normalise them,
which lambdas weight the scores) are kept, INCLUDING the cross-wired statistics of the two-bank classes
(every std comes from the rgb library and every mean from the xyz library,
multiple_features.py:372-377, 877-880).  Banks and patches stay on the GPU; extraction and scoring are
the HIP kernels behind ``Features``.
"""
import math
import os
import warnings

import numpy as np
import torch

from .. import engine as eng
from .. import ops
from .features import Features


def organized_pc_to_unorganized_pc_no_zeros(sample):
    """multiple_features.py:10-25: sample[1] [1,3,H,W] -> (pc [1,3,N] without all-zero-coordinate pixels,
    nonzero_indices [N]).  Host numpy, exactly as the reference (a1 is negligible at B = 1; the batched
    engine uses the cmdiad_unorganize kernel instead)."""
    pc = sample[1].squeeze().permute(1, 2, 0).numpy()
    flat = pc.reshape(pc.shape[0] * pc.shape[1], pc.shape[2])
    nz = np.nonzero(np.all(flat != 0, axis=1))[0]
    return torch.tensor(flat[nz, :]).unsqueeze(dim=0).permute(0, 2, 1), nz


def _side(patch):
    s = int(math.sqrt(patch.shape[0]))
    return (s, s)


def _score_samples(fuser, x):
    """`fuser.score_samples(x)` for a fitted scikit-learn SGDOneClassSVM without its per-call input validation (1.4 ms per 50 176 x 2
    map against 0.1 ms for the arithmetic): the same numpy operations in the same order as
    SGDOneClassSVM.decision_function / score_samples -- (X @ coef_.T - offset_).ravel() + offset_ -- so the same bits
    (tests/test_host_cpu.py::test_fast_score_samples_is_sklearns).  Anything else goes through the object's own method."""
    from sklearn.linear_model import SGDOneClassSVM
    if type(fuser) is SGDOneClassSVM and isinstance(x, np.ndarray) and x.ndim == 2 and x.dtype in (np.float32, np.float64) \
            and hasattr(fuser, "coef_") and x.shape[1] == fuser.coef_.shape[-1] and np.isfinite(x).all():
        return (x @ fuser.coef_.T - fuser.offset_).ravel() + fuser.offset_
    return fuser.score_samples(x)


class _PendingScores:
    """Scores of a micro-batch whose device work and device->host copy are queued but not yet waited for: `result()` (also
    indexing / iteration) waits and builds the per-sample (s [1,k], s_map [gt*gt,k]) pairs.  _MethodBase._flush keeps ONE such batch in
    flight, so the host-side work of a batch (sklearn's score_samples, the result lists) runs beside the next batch's GPU work."""

    def __init__(self, host, event, keep, B, k, gt, lam_s, lam_map):
        self._host, self._event, self._keep = host, event, keep
        self._B, self._k, self._gt, self._lam_s, self._lam_map = B, k, gt, lam_s, lam_map
        self._out = None

    def result(self):
        if self._out is None:
            self._event.synchronize()
            host, B, k, gt = self._host.numpy(), self._B, self._k, self._gt
            out = []
            for b in range(B):
                # np.float32 x np.float32: the reference's `lambda * tensor` is a float32 product (explicit, so the bits do not
                # depend on NumPy's scalar promotion rules: NumPy < 2 would compute python-float x float32 in float64)
                s = torch.tensor([[float(np.float32(self._lam_s[i]) * host[B * k * gt * gt + b * k + i]) for i in range(k)]])
                # numpy for the host-side products: torch's CPU intra-op pool (one thread per core by default) stalls for
                # 80 ms every few calls on 50 176-element tensors on a 128-core host; same float32 arithmetic either way
                cols = [np.float32(self._lam_map[i]) * host[(b * k + i) * gt * gt:(b * k + i + 1) * gt * gt] for i in range(k)]
                out.append((s, torch.from_numpy(np.stack(cols, axis=1))))
            self._out, self._keep = out, None
        return self._out

    def __getitem__(self, i):
        return self.result()[i]

    def __iter__(self):
        return iter(self.result())

    def __len__(self):
        return self._B


def _lazy_result(name, kind):
    """Result attribute of the reference (a plain list there) that FLUSHES the deferred micro-batch of its phase before it is
    read, so an observer sees exactly what the reference's eager loop would have appended by then."""
    key = "_lz_" + name

    def get(self):
        if not self.__dict__.get("_flushing", False):
            try:
                self._flush(kind)
            except AttributeError as e:
                # nn.Module.__getattr__ swallows an AttributeError raised inside a property getter and reports
                # "object has no attribute <name>" instead: keep the real cause visible
                raise RuntimeError(f"running the deferred '{kind}' micro-batch behind .{name} failed: {e!r}") from e
        return self.__dict__[key]

    def set_(self, value):
        self.__dict__[key] = value

    return property(get, set_)


class _MethodBase(Features):
    """Shared plumbing of the method classes (not part of the reference's public surface).

    The five protocol calls keep the reference's signatures and order of effects, but the per-sample calls
    (add_sample_to_mem_bank / add_sample_to_late_fusion_mem_bank / predict) only QUEUE the sample: the queue is run through
    the batched engine every CMDIAD_PREDICT_BATCH samples (default 16), at the next phase call (run_coreset /
    run_late_fusion / calculate_metrics) and whenever a result attribute is read.  The reference itself reads those results
    only at the phase boundaries (cmdiad_runner.py:44-92), every sample is independent of the others, and a
    sample's numbers do not depend on the batch it rode in (tests/test_gpu_engine.py::test_batch_invariance,
    test_gpu_predictor.py::test_dropin_micro_batching_is_invisible); at B = 1 the GPU is bound by a 2 ms single-CU FPS
    chain and ~200 tiny launches per image, in a micro-batch of 16 those run side by side."""
    _image_slot = 0      # which element of a sample goes through the RGB backbone (DepthFeatures: 2)


    patch_xyz_lib = _lazy_result("patch_xyz_lib", "fit")
    patch_rgb_lib = _lazy_result("patch_rgb_lib", "fit")
    patch_fusion_lib = _lazy_result("patch_fusion_lib", "fit")
    s_lib = _lazy_result("s_lib", "late")
    s_map_lib = _lazy_result("s_map_lib", "late")
    image_preds = _lazy_result("image_preds", "predict")
    image_labels = _lazy_result("image_labels", "predict")
    pixel_preds = _lazy_result("pixel_preds", "predict")
    pixel_labels = _lazy_result("pixel_labels", "predict")
    predictions = _lazy_result("predictions", "predict")
    gts = _lazy_result("gts", "predict")
    img_name = _lazy_result("img_name", "predict")

    # ------------------------------------------------------------------ deferred micro-batches
    def _micro_batch(self):
        return max(1, int(os.environ.get("CMDIAD_PREDICT_BATCH", "16")))

    def _defer(self, kind, item):
        q = self.__dict__.setdefault("_pending", {"fit": [], "late": [], "predict": []})
        q[kind].append(item)
        if len(q[kind]) >= self._micro_batch():
            self._flush(kind, drain=False)

    def _flush(self, kind, drain=True):
        """Run the queued samples of a phase.  Scoring phases keep one micro-batch IN FLIGHT: its device work is queued, then the
        previous batch is completed on the host (score_samples, result lists) while the GPU works; `drain` (every read of a result
        attribute and every phase call) completes the batch just queued as well, so an observer never sees a partial list.

        A sample that cannot be processed (e.g. a cloud with fewer points than the grouping needs) must not take the rest of its
        micro-batch with it: when a batch fails its samples are run ONE BY ONE in call order -- the valid ones before the bad
        one are recorded, the bad one raises (as its own call would have in the reference's eager loop), and the ones behind it
        go back to the head of the queue for the next flush."""
        q = self.__dict__.get("_pending")
        flight = self.__dict__.setdefault("_inflight", {})
        if (not q or not q[kind]) and kind not in flight:
            return
        items = []
        if q and q[kind]:
            items, q[kind] = q[kind], []
        self.__dict__["_flushing"] = True
        try:
            try:
                self._run_items(kind, items, drain, flight)
            except Exception as exc:
                # only a failure of the NEW samples' batch is retried sample by sample; anything else (completing the batch that was
                # in flight, recording results) is not tied to one sample and must surface as it is
                if len(items) <= 1 or not getattr(exc, "_cmdiad_new_batch", False):
                    raise
                for k, it in enumerate(items):
                    try:
                        self._run_items(kind, [it], True, flight)
                    except Exception:
                        q[kind] = items[k + 1:] + q[kind]
                        raise
                # every sample went through on its own: the results are complete, but the batch-level failure is worth knowing
                warnings.warn(f"a micro-batch of {len(items)} '{kind}' samples failed as a batch ({exc!r}) and succeeded sample by sample")
        finally:
            self.__dict__["_flushing"] = False

    def _run_items(self, kind, items, drain, flight):
        def new_batch(fn, *a, **k):
            try:
                return fn(*a, **k)
            except Exception as exc:
                exc._cmdiad_new_batch = True       # nothing of this call has been recorded yet: safe to retry per sample
                raise
        if kind == "fit":
            if items:
                new_batch(self._fit_batch, items)
            return
        new = None
        if items:
            new = (new_batch(self._score_batch, items if kind == "late" else [it[0] for it in items], test=(kind == "predict")), items)
        if kind in flight:
            self._complete(kind, *flight.pop(kind))
        if new is not None:
            if drain:
                self._complete(kind, *new)
            else:
                flight[kind] = new

    def _complete(self, kind, scores, items):
        if kind == "late":
            for s, s_map in scores:
                self.s_lib.append(s)
                self.s_map_lib.append(s_map)
        else:
            for (s, s_map), (_, mask, label, rgb_path) in zip(scores, items):
                self._record(s, s_map, mask, label, rgb_path)

    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.class_name = class_name
        self._defer("fit", sample)

    def add_sample_to_late_fusion_mem_bank(self, sample):
        self._defer("late", sample)

    def predict(self, sample, mask, label, rgb_path):
        self._defer("predict", (sample, mask, label, rgb_path))

    def run_late_fusion(self):
        self._flush("late")
        super().run_late_fusion()

    def calculate_metrics(self):
        self._flush("predict")
        super().calculate_metrics()

    # ------------------------------------------------------------------ batched extraction
    def _extract_batch(self, samples, want_rgb=True, want_xyz=True):
        """Device-resident extraction of a micro-batch (engine.Engine.extract; the zero-pixel compaction of
        organized_pc_to_unorganized_pc_no_zeros runs in cmdiad_unorganize, bit-identical order)."""
        dev = self.device

        def staged(tensors):
            # gathered straight into pinned memory and copied asynchronously: a copy from pageable memory makes the host wait for
            # everything queued before it, i.e. for the micro-batch still in flight (_flush), and the GPU then waits for the host
            host = torch.empty((len(tensors), *tensors[0].shape[1:]), dtype=torch.float32, pin_memory=True)
            if all(t.dtype == torch.float32 and t.device.type == "cpu" for t in tensors):
                torch.cat(tensors, out=host)
            else:
                host.copy_(torch.cat([t.cpu() for t in tensors]))
            return host, host.to(dev, non_blocking=True)

        rgb = staged([s[self._image_slot] for s in samples])[1] if want_rgb else None
        if not want_xyz:
            with torch.no_grad():
                return self._engine.extract(rgb, want_xyz=False)
        pcs, pcs_dev = staged([s[1] for s in samples])
        flat = pcs.numpy().reshape(len(samples), 3, -1)   # numpy: no 150 k-element op through torch's CPU thread pool
        counts = np.count_nonzero(np.all(flat != 0, axis=1), axis=1)
        if counts.min() < self.args.group_size:
            raise ValueError(f"point cloud has {int(counts.min())} valid points; the {self.args.group_size}-nearest-neighbour "
                             f"grouping (models/models.py:88-113) needs at least {self.args.group_size}")
        side = self.__dict__.get("_side_stream")
        if side is None:
            side = self.__dict__["_side_stream"] = ops.shared_stream(dev, "predictor.side")
        with torch.no_grad():
            return self._engine.extract(rgb, pcs_dev, want_rgb=want_rgb, n_max=int(counts.max()),
                                        side_stream=side if want_rgb else None)

    def _coreset(self, lib, name):
        if self.f_coreset < 1:
            idx = self.get_coreset_idx_randomp(lib, n=int(self.f_coreset * lib.shape[0]), eps=self.coreset_eps,
                                               lib=name, coreset_dtype=self.coreset_dtype)
            self.coreset_idx = idx
            return lib[idx.to(lib.device)]
        return lib

    def _score_columns(self, columns):
        """columns: [(patch [B,Q,D] raw, mean, std, modal, lambda_s, lambda_map)] in the reference's column order ->
        per sample (s [1,k], s_map [gt*gt, k]) exactly as the reference stacks them (e.g. multiple_features.py:985-992).
        The device work of every column is queued first, then ONE blur launch (a block per map) and ONE asynchronous
        device->host copy: the return value is a _PendingScores (list-like; waits when first read)."""
        from .. import ops
        gt = self.gt_size
        # The scoring of a micro-batch (two library searches, exact re-score, re-weighting, maps, blur: ~3 ms of whole-chip work for
        # 16 samples) runs on the post stream, BESIDE the extraction of the next micro-batch, whose launches the caller queues on the
        # current stream as soon as this function returns -- what the batched predictor does with its HIP graphs, here with eager
        # launches.  The patch tensors were produced on the current stream: the post stream waits for it, and the caching allocator
        # is told that they are read there (record_stream).  CMDIAD_DROPIN_POST=0: everything on the current stream (rounds 1-5).
        dev = torch.device(self.device)
        cur = torch.cuda.current_stream(dev)
        post = ops.shared_stream(dev, "predictor.post") if os.environ.get("CMDIAD_DROPIN_POST", "1") != "0" else cur
        columns = [(c[0].to(dev), *c[1:]) for c in columns]
        banks = [self._bank(getattr(self, f"patch_{c[3]}_lib")) for c in columns]   # (built on first use: on the CURRENT stream)
        if post is not cur:
            post.wait_stream(cur)
            for c in columns:
                c[0].record_stream(post)
        with torch.cuda.stream(post):
            rs = []
            for (patch, mean, std, modal, _, _), bank in zip(columns, banks):
                q = eng.normalize(patch.float().contiguous(), mean, std)        # a11: fused HIP kernel
                side = int(math.sqrt(q.shape[1]))
                rs.append(eng.score_patches(q, bank, (side, side), gt))
            B, k = columns[0][0].shape[0], len(columns)
            maps = ops.blur8_maps(torch.stack([r["s_map_pre"] for r in rs], 1).reshape(B * k, gt, gt).contiguous(),
                                  float(self.blur.radius))
            dev_out = torch.cat([maps.reshape(-1), torch.stack([r["s"] for r in rs], 1).reshape(-1).float()])
            host = torch.empty(dev_out.shape, dtype=dev_out.dtype, pin_memory=True)
            host.copy_(dev_out, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        return _PendingScores(host, event, dev_out, B, k, gt, [c[4] for c in columns], [c[5] for c in columns])

    def _record(self, s, s_map, mask, label, rgb_path):
        # from_numpy, not torch.tensor(): no 50 176-element copy through torch's CPU thread pool (see _score_columns)
        s = torch.from_numpy(np.ascontiguousarray(_score_samples(self.detect_fuser, s.numpy())))
        s_map = torch.from_numpy(np.ascontiguousarray(_score_samples(self.seg_fuser, s_map.numpy()))).view(1, self.gt_size, self.gt_size)
        self.image_preds.append(s.numpy())
        self.image_labels.append(label)
        self.pixel_preds.extend(s_map.flatten().numpy())
        self.pixel_labels.extend(mask.flatten().numpy())
        self.predictions.append(s_map.detach().cpu().squeeze().numpy())
        self.gts.append(mask.detach().cpu().squeeze().numpy())
        self.img_name.append(rgb_path)
        if getattr(self.args, "save_seg_results", False):
            path = rgb_path[0].replace('mvtec_3d', 'segmentation').replace('png', 'pt')
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(s_map, path)


class RGBFeatures(_MethodBase):
    def _fit_batch(self, samples):
        ex = self._extract_batch(samples, want_xyz=False)
        self.patch_rgb_lib.extend(eng.Engine.rgb_patch(ex).unbind(0))

    def run_coreset(self):
        self._flush("fit")
        self.patch_rgb_lib = torch.cat(self.patch_rgb_lib, 0)
        self.rgb_mean, self.rgb_std = torch.mean(self.patch_rgb_lib), torch.std(self.patch_rgb_lib)
        self.patch_rgb_lib = self._coreset(eng.normalize(self.patch_rgb_lib, self.rgb_mean, self.rgb_std), 'patch_rgb_lib')

    def _score_batch(self, samples, test=False):
        ex = self._extract_batch(samples, want_xyz=False)
        return self._score_columns([(eng.Engine.rgb_patch(ex), self.rgb_mean, self.rgb_std, 'rgb',
                                     self.args.rgb_s_lambda, self.args.rgb_smap_lambda)])


class DepthFeatures(RGBFeatures):
    """multiple_features.py:124-200: the RGB method fed with the three-channel depth image of the sample (sample[2]) instead of
    the photograph -- same backbone, same library, same scoring.  (No caller in the reference constructs it: cmdiad_runner.py:16-31.)"""
    _image_slot = 2


class PointFeatures(_MethodBase):
    def _fit_batch(self, samples):
        ex = self._extract_batch(samples, want_rgb=False)
        self.patch_xyz_lib.extend(self._engine.xyz_patch(ex, P=56).unbind(0))

    def run_coreset(self):
        self._flush("fit")
        self.patch_xyz_lib = torch.cat(self.patch_xyz_lib, 0)
        self.xyz_mean, self.xyz_std = torch.mean(self.patch_xyz_lib), torch.std(self.patch_xyz_lib)
        self.patch_xyz_lib = self._coreset(eng.normalize(self.patch_xyz_lib, self.xyz_mean, self.xyz_std), 'patch_xyz_lib')

    def _score_batch(self, samples, test=False):
        ex = self._extract_batch(samples, want_rgb=False)
        return self._score_columns([(self._engine.xyz_patch(ex, P=56), self.xyz_mean, self.xyz_std, 'xyz',
                                     self.args.xyz_s_lambda, self.args.xyz_smap_lambda)])


class DoubleRGBPointFeatures(_MethodBase):
    def _save_features(self, xyz_patch, rgb_patch2, split):
        # the trainer's on-disk format: [3136, 768 xyz | 768 rgb] f32 per sample (multiple_features.py:815-825, 942-945)
        for sub in ("", "train", "test"):
            os.makedirs(os.path.join(self.args.save_path, sub), exist_ok=True)
        for x, r in zip(xyz_patch, rgb_patch2):
            torch.save(torch.cat([x, r], dim=1).cpu(), os.path.join(self.args.save_path, split, self.class_name + str(self.ins_id) + '.pt'))
            self.ins_id += 1

    def _save_pairs(self, samples, ex, xyz_patch, rgb_patch2, split):
        """--save_frgb_xyz / --save_rgb_fxyz (multiple_features.py:827-867 while the memory bank is built, :947-962 in predict):
        the training pairs of the feature-to-input and input-to-feature heads, per sample
          <save_path_frgb_xyz>/<split>/frgb/<class><i>_frgb.pt  [3136, 768] f32 (ViT features on the 56 x 56 grid)
          <save_path_frgb_xyz>/<split>/xyz/<class><i>_xyz.pt    [3, 224, 224]     (the organised point map as given)
          <save_path_rgb_fxyz>/<split>/fxyz/<class><i>_hfxyz.pt [3136, 768] f32, _lfxyz.pt [784, 768] f32 (Point-MAE patch features
                                                                 pooled to 56 x 56 / 28 x 28), <split>/rgb/<class><i>_rgb.pt [3, 224, 224]
        read back by cmdiad_amd.dataset.{FeatureToInput,InputToFeature}PreTrainTensorDataset.  Tensors are saved from the host,
        where the reference's patches live (features.py:139-140)."""
        a = self.args
        if getattr(a, "save_frgb_xyz", False):
            for sub in ("frgb", "xyz"):
                for sp in ("train", "test"):
                    os.makedirs(os.path.join(a.save_path_frgb_xyz, sp, sub), exist_ok=True)
            for smp, r in zip(samples, rgb_patch2):
                pc = smp[1].squeeze()
                assert tuple(pc.shape) == (3, 224, 224)
                stem = self.class_name + str(self.ins_id2)
                torch.save(r.cpu(), os.path.join(a.save_path_frgb_xyz, split, 'frgb', stem + '_frgb.pt'))
                torch.save(pc.cpu(), os.path.join(a.save_path_frgb_xyz, split, 'xyz', stem + '_xyz.pt'))
                self.ins_id2 += 1
        if getattr(a, "save_rgb_fxyz", False):
            for sub in ("rgb", "fxyz"):
                for sp in ("train", "test"):
                    os.makedirs(os.path.join(a.save_path_rgb_fxyz, sp, sub), exist_ok=True)
            low = self._engine.xyz_patch(ex, P=28)          # get_xyz_patch(..., get_2828=True), features.py:169-184
            for smp, hi, lo in zip(samples, xyz_patch, low):
                img = smp[0].squeeze()
                assert tuple(lo.shape) == (784, 768) and tuple(hi.shape) == (3136, 768) and tuple(img.shape) == (3, 224, 224)
                stem = self.class_name + str(self.ins_id3)
                torch.save(hi.cpu(), os.path.join(a.save_path_rgb_fxyz, split, 'fxyz', stem + '_hfxyz.pt'))
                torch.save(lo.cpu(), os.path.join(a.save_path_rgb_fxyz, split, 'fxyz', stem + '_lfxyz.pt'))
                torch.save(img.cpu(), os.path.join(a.save_path_rgb_fxyz, split, 'rgb', stem + '_rgb.pt'))
                self.ins_id3 += 1

    def _patches(self, samples, split=None):
        if getattr(self.args, "use_depth", False) and split != 'train':
            samples = [(s[1], s[1], *s[2:]) for s in samples]   # multiple_features.py:931-932: the point map stands in for the image
        ex = self._extract_batch(samples)
        xyz_patch, rgb_patch, rgb_patch2 = self._engine.xyz_patch(ex, P=56), eng.Engine.rgb_patch(ex), eng.Engine.rgb_patch56(ex)
        if split is not None:
            if getattr(self.args, "save_feature_for_fusion", False):
                self._save_features(xyz_patch, rgb_patch2, split)
            if getattr(self.args, "save_frgb_xyz", False) or getattr(self.args, "save_rgb_fxyz", False):
                self._save_pairs(samples, ex, xyz_patch, rgb_patch2, split)
        return xyz_patch, rgb_patch, rgb_patch2

    def _fit_batch(self, samples):
        xyz_patch, rgb_patch, _ = self._patches(samples, 'train')
        self.patch_xyz_lib.extend(xyz_patch.unbind(0))
        self.patch_rgb_lib.extend(rgb_patch.unbind(0))

    def run_coreset(self):
        self._flush("fit")
        self.patch_xyz_lib = torch.cat(self.patch_xyz_lib, 0)
        self.patch_rgb_lib = torch.cat(self.patch_rgb_lib, 0)
        # cross-wired exactly as the reference (multiple_features.py:877-880)
        self.xyz_mean = torch.mean(self.patch_xyz_lib)
        self.xyz_std = torch.std(self.patch_rgb_lib)
        self.rgb_mean = torch.mean(self.patch_xyz_lib)
        self.rgb_std = torch.std(self.patch_rgb_lib)
        self.patch_xyz_lib = self._coreset(eng.normalize(self.patch_xyz_lib, self.xyz_mean, self.xyz_std), 'patch_xyz_lib')
        self.patch_rgb_lib = self._coreset(eng.normalize(self.patch_rgb_lib, self.rgb_mean, self.rgb_std), 'patch_rgb_lib')

    def _score_batch(self, samples, test=False):
        xyz_patch, rgb_patch, rgb_patch2 = self._patches(samples, 'test' if test else None)
        a = self.args
        return self._score_columns([(xyz_patch, self.xyz_mean, self.xyz_std, 'xyz', a.xyz_s_lambda, a.xyz_smap_lambda),
                                    (rgb_patch, self.rgb_mean, self.rgb_std, 'rgb', a.rgb_s_lambda, a.rgb_smap_lambda)])


class RGBorXYZWithOneHallucination(_MethodBase):
    """MTFI with one real and one hallucinated modality (multiple_features.py:312-573): the main modality's real features
    plus the other modality's features hallucinated either from the main modality's FEATURES (``--use_hn``: the FtoF MLP, or
    the FtoF conv head when ``--use_hn_conv`` is given as well) or from the main modality's INPUT (``--use_hrnet``, ItoF)."""

    def _hallucinate(self, samples, xyz_patch, rgb_patch2):
        """-> [B, 3136, 768] hallucinated features of the OTHER modality."""
        a = self.args
        if a.main_modality not in ('rgb', 'xyz'):
            raise Exception('Unknown modality')
        with torch.no_grad():
            if getattr(a, "use_hrnet", False):  # multiple_features.py:326-331, 343-348: from the raw image / point map
                src = torch.cat([s[0] if a.main_modality == 'rgb' else s[1] for s in samples])
                h = self.fusion.hallucination_tokens(src.to(self.device))
                assert tuple(h.shape[1:]) == (3136, 768)
            elif a.main_modality == 'rgb':
                h = self.fusion.hallucination_generation(rgb_feature=rgb_patch2, out_type='xyz')
            else:
                h = self.fusion.hallucination_generation(xyz_feature=xyz_patch, out_type='rgb')
        assert len(h.shape) == 3
        return h.detach()

    def _patches(self, samples, fit=False):
        """-> (xyz_patch [B,3136,768] | None, rgb_patch [B,784,768] | None, hallucination [B,3136,768])."""
        ex = self._extract_batch(samples)
        xyz_patch = self._engine.xyz_patch(ex, P=56)
        return xyz_patch, eng.Engine.rgb_patch(ex), self._hallucinate(samples, xyz_patch, eng.Engine.rgb_patch56(ex))

    def _fit_batch(self, samples):
        xyz_patch, rgb_patch, hall = self._patches(samples, fit=True)
        self.patch_rgb_lib.extend(rgb_patch.unbind(0))
        self.patch_xyz_lib.extend(xyz_patch.unbind(0))
        self.patch_fusion_lib.extend(hall.unbind(0))

    def run_coreset(self):
        self._flush("fit")
        self.patch_xyz_lib = torch.cat(self.patch_xyz_lib, 0)
        self.patch_rgb_lib = torch.cat(self.patch_rgb_lib, 0)
        self.patch_fusion_lib = torch.cat(self.patch_fusion_lib, 0)
        # multiple_features.py:372-377: means from the xyz library, stds from the rgb library
        self.xyz_mean = self.rgb_mean = self.fusion_mean = torch.mean(self.patch_xyz_lib)
        self.xyz_std = self.rgb_std = self.fusion_std = torch.std(self.patch_rgb_lib)
        if self.args.main_modality == 'rgb':
            self.patch_rgb_lib = self._coreset(eng.normalize(self.patch_rgb_lib, self.rgb_mean, self.rgb_std), 'patch_rgb_lib')
        elif self.args.main_modality == 'xyz':
            self.patch_xyz_lib = self._coreset(eng.normalize(self.patch_xyz_lib, self.xyz_mean, self.xyz_std), 'patch_xyz_lib')
        self.patch_fusion_lib = self._coreset(eng.normalize(self.patch_fusion_lib, self.fusion_mean, self.fusion_std),
                                              'patch_fusion_lib')

    def _score_batch(self, samples, test=False):
        xyz_patch, rgb_patch, hall = self._patches(samples)
        a = self.args
        if a.main_modality == 'rgb':
            main = (rgb_patch, self.rgb_mean, self.rgb_std, 'rgb', a.rgb_s_lambda, a.rgb_smap_lambda)
        else:
            main = (xyz_patch, self.xyz_mean, self.xyz_std, 'xyz', a.xyz_s_lambda, a.xyz_smap_lambda)
        # the reference scores the hallucinated library first (:514-518) and stacks [main, fusion] (:525,533)
        fus = (hall, self.fusion_mean, self.fusion_std, 'fusion', a.fusion_s_lambda, a.fusion_smap_lambda)
        return self._score_columns([main, fus])


class RGBorXYZWithOneHallucinationFromFeature(RGBorXYZWithOneHallucination):
    """MTFI feature-to-INPUT (multiple_features.py:576-797; ``--use_hn_from_rgb_mlp`` / ``--use_hn_from_rgb_conv``): the head
    turns the main modality's features into the OTHER modality's input -- an organised point map [1,3,224,224] from rgb
    features, or an RGB image from xyz features -- and the frozen extractor of that modality is run on it; those
    re-extracted features are the "hallucination" library / query.  Banks, statistics, coreset and scoring are
    the parent's (the reference repeats them verbatim, :614-648, :754-797).

    As in the reference, with main_modality == 'rgb' the real point cloud is only read while the memory bank is built
    (its patches feed the cross-wired statistics, :582,605,612-618); late fusion and predict never touch it (:651-663,
    :701-719), so the 3-D branch of the extractor is skipped there."""

    def _patches(self, samples, fit=False):
        a = self.args
        B = len(samples)
        if a.main_modality == 'rgb':
            ex = self._extract_batch(samples, want_rgb=True, want_xyz=fit)
            rgb_patch, rgb_patch2 = eng.Engine.rgb_patch(ex), eng.Engine.rgb_patch56(ex)
            xyz_patch = self._engine.xyz_patch(ex, P=56) if fit else None
            with torch.no_grad():
                pc = self.fusion.hallucination_generation(rgb_patch2)  # [B,3,224,224] hallucinated point maps
            assert tuple(pc.shape) == (B, 3, self.xyz_size, self.xyz_size), tuple(pc.shape)
            # zero-coordinate pixels dropped as :592-594; hallucinated maps have their own point counts: one by one
            hall = torch.cat([self._engine.xyz_patch(self._extract_device(None, pc[b:b + 1], want_rgb=False, want_xyz=True), P=56)
                              for b in range(B)])
        elif a.main_modality == 'xyz':
            ex = self._extract_batch(samples, want_rgb=fit, want_xyz=True)
            xyz_patch = self._engine.xyz_patch(ex, P=56)
            rgb_patch = eng.Engine.rgb_patch(ex) if fit else None
            with torch.no_grad():
                img = self.fusion.hallucination_generation(xyz_patch)
            assert tuple(img.shape) == (B, *samples[0][0].shape[1:]), (tuple(img.shape), tuple(samples[0][0].shape))
            with torch.no_grad():
                hall = eng.Engine.rgb_patch(self._engine.extract(img.to(self.device, torch.float32), want_xyz=False))
        else:
            raise NotImplementedError
        return xyz_patch, rgb_patch, hall
