"""Drop-in for the reference's ``feature_extractors/features.py`` (class ``Features``, lines 21-425):
same constructor, attributes and method names; extraction, patching and kNN scoring run in the HIP
kernels through cmdiad_amd.engine.

Differences a caller can observe (also listed in INTEGRATION.md):
  * patch tensors returned by get_rgb_patch / get_xyz_patch and the memory banks live on the GPU
    (the reference moves everything to the CPU, features.py:137-140, and then runs cdist there);
  * ``interpolated_feature_maps`` is a lazy handle (``LazyInterpolated``): the 154 MB [1,768,N] tensor
    is only materialised when ``.materialize()`` / ``.to()`` is called; get_xyz_patch consumes the handle;
  * ``calculate_dist`` returns a ``DistHandle`` (the Q x N matrix is never built; ``.materialize()`` builds it exactly,
    ``.min(1)`` gives torch.min(dist, 1) without it);
  * the feature extractor runs in eval mode, dist_method_s must be 'l2' (the reference's
    l1 / cos branches are broken).
"""
import math
import os

import numpy as np
import torch
from sklearn import linear_model, random_projection
from sklearn.metrics import roc_auc_score

from .. import engine as eng
from .. import ops
from ..models.models import Model
from ..utils.utils import KNNGaussianBlur, set_seeds


class LazyInterpolated:
    """Stands in for ``interpolating_points(...)`` of features.py:137: holds the 3-NN indices / weights and
    the centre features on the device; ``materialize()`` produces the reference's [B, D, N] tensor."""

    def __init__(self, extraction):
        self.ex = extraction

    @property
    def shape(self):
        B, G, D = self.ex.xyz_feats.shape
        return torch.Size((B, D, self.ex.idx3.shape[1]))

    def materialize(self):
        return ops.interp_gather(self.ex.xyz_feats, self.ex.idx3, self.ex.w3, self.ex.n_valid).permute(0, 2, 1)

    def to(self, *a, **k):
        return self.materialize().to(*a, **k)


class DistHandle:
    """Stands in for the Q x N distance matrix of features.py:190 (never built on the product path)."""

    def __init__(self, patch, lib):
        self.patch, self.lib = patch, lib

    @property
    def shape(self):
        return torch.Size((self.patch.shape[0], self.lib.shape[0]))

    def materialize(self):
        """The reference's Q x N matrix (features.py:190), exact fp32 (sum of squared differences, cmdiad_l2_dist_matrix):
        960 MB at the reference's sizes -- API compatibility and debugging only, the scoring path never builds it."""
        dev = self.lib.device if self.lib.is_cuda else "cuda"
        return ops.l2_dist_matrix(self.patch.to(dev).float().contiguous(), self.lib.to(dev).float().contiguous())

    def min(self, dim=1):
        """torch.min(dist, dim=1) of the reference (features.py:227) without the matrix: (values, indices)."""
        if dim != 1:
            raise NotImplementedError("DistHandle.min: dim=1 only")
        dev = self.lib.device if self.lib.is_cuda else "cuda"
        bank = eng.Bank(self.lib.to(dev).float())
        q = self.patch.to(dev).float().contiguous()
        q16, _, qsq = ops.normalize_cast(q)
        keys = ops.l2_min_keys(q16, qsq, bank.bf16, bank.sqnorm, ops.new_keys(q.shape[0], q.device, runner=True))
        return ops.l2_rescore(q, bank.f32, keys)      # best and runner-up measured in fp32: torch.min's answer on near-ties too


class PixelList:
    """List-like store for ``pixel_preds`` / ``pixel_labels`` (features.py:78-79 keeps them as Python lists and extends them
    by 50 176 numpy scalars per image, features.py:305 turns them into an array at the end).  Same protocol -- extend,
    len, iteration, indexing, ``np.array(x)`` -- but the values stay in per-image numpy chunks: millions of boxed floats in
    one list made the cyclic garbage collector stall the predict loop for 60-90 ms every few images."""

    def __init__(self):
        self._chunks, self._flat = [], None

    def extend(self, values):
        self._chunks.append(np.asarray(values).reshape(-1))
        self._flat = None

    def append(self, value):
        self.extend([value])

    def __array__(self, dtype=None, copy=None):
        if self._flat is None:
            self._flat = np.concatenate(self._chunks) if self._chunks else np.zeros((0,))
            self._chunks = [self._flat]
        return self._flat if dtype is None else self._flat.astype(dtype, copy=False)

    def __len__(self):
        return sum(c.shape[0] for c in self._chunks)

    def __iter__(self):
        return iter(self.__array__())

    def __getitem__(self, i):
        return self.__array__()[i]


class _NeedsFit(Exception):
    pass


class Features(torch.nn.Module):
    def __init__(self, args, image_size=224, f_coreset=0.1, coreset_eps=0.9, shared_extractor=None):
        """shared_extractor (not in the reference's signature): an already constructed ``Model`` to use instead of building and
        loading the frozen backbones again -- the reference's class loop builds a fresh object, i.e. re-loads both checkpoints,
        for every class (main.py:23); cmdiad_amd.evaluate hands the rank's extractor from class to class."""
        super().__init__()
        from .. import _native
        _native.lib()  # fail loudly at construction when libcmdiad_hip.so is missing
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        if shared_extractor is not None:
            self.deep_feature_extractor = shared_extractor
        else:
            self.deep_feature_extractor = Model(
                device=self.device, rgb_backbone_name=args.rgb_backbone_name, xyz_backbone_name=args.xyz_backbone_name,
                group_size=args.group_size, num_group=args.num_group,
                checkpoint_path=getattr(args, "rgb_checkpoint_path", "") or "")   # offline stand-in for timm's hub download
            self.deep_feature_extractor.to(self.device)
            self.deep_feature_extractor.eval()

        self.args = args
        self.class_name = None
        self.rgb_size, self.xyz_size, self.gt_size = args.rgb_size, args.xyz_size, args.gt_size
        self.f_coreset, self.coreset_eps, self.coreset_dtype = args.f_coreset, args.coreset_eps, args.coreset_dtype
        self.blur = KNNGaussianBlur(4)
        self.n_reweight = 3
        set_seeds(0)
        self.patch_xyz_lib, self.patch_rgb_lib, self.patch_fusion_lib = [], [], []
        self.patch_lib, self.patch_share_lib, self.patch_non_share_lib = [], [], []
        self.random_state = args.random_state
        self.xyz_dim = self.rgb_dim = 0
        self.xyz_mean = self.xyz_std = self.rgb_mean = self.rgb_std = self.fusion_mean = self.fusion_std = 0
        self.share_mean = self.share_std = self.non_share_mean = self.non_share_std = 0
        self.image_preds, self.image_labels, self.pixel_preds, self.pixel_labels = [], [], PixelList(), PixelList()
        self.gts, self.predictions = [], []
        self.image_rocauc = self.pixel_rocauc = self.au_pro = self.au_pro_001 = 0
        self.ins_id = self.ins_id2 = self.ins_id3 = 0

        # features.py:91-105.  Every head goes to the GPU (the reference forgets .cuda() for two of them and then feeds them
        # CUDA tensors); later flags override earlier ones, as there.
        from ..models import hallucination_network as hn
        if getattr(args, "use_hn", False):
            self.fusion = hn.HallucinationCrossModalityNetwork(args, 768, 768, hidden_ratio=2.5)
        if getattr(args, "use_hn_conv", False):
            self.fusion = hn.HallucinationCrossModalityConv(args, 768, 768)
        if getattr(args, "use_hn_from_rgb_mlp", False):
            self.fusion = hn.HallucinationRGBFeatureToXYZInputMLP(args, 768)
        if getattr(args, "use_hn_from_rgb_conv", False):
            self.fusion = hn.HallucinationFeatureToInputConv(args, 768)
        if getattr(args, "use_hrnet", False):
            from ..models.hrnet import HRNet
            self.fusion = HRNet(args.c_hrnet, 768, 0.1)
        if getattr(self, "fusion", None) is not None:
            self.fusion.to(self.device)
        if getattr(args, "use_uff", False):
            raise NotImplementedError("--use_uff calls fusion.feature_fusion(), which no module of the reference defines "
                                      "(multiple_features.py:324; a leftover of M3DM's UFF)")
        if getattr(args, "fusion_module_path", "") != "":
            ckpt = torch.load(args.fusion_module_path, map_location="cpu")["model"]
            print("[Fusion Block]", self.fusion.load_state_dict(ckpt))
            self.fusion.eval()

        # features.py:127-128.  CMDIAD_OCSVM_DEVICE=1 fits them on the GPU (cmdiad_ocsvm_fit: scikit-learn's float32 SGD in the same
        # update order, identical coef_ / offset_ / n_iter_, tests/test_gpu_ocsvm.py); the default stays scikit-learn on the host,
        # which is faster at this strictly sequential recurrence (docs/history.md section 7)
        if os.environ.get("CMDIAD_OCSVM_DEVICE", "0") == "1":
            from ..ocsvm import DeviceSGDOneClassSVM as _OCSVM
        else:
            _OCSVM = linear_model.SGDOneClassSVM
        self.detect_fuser = _OCSVM(random_state=42, nu=args.ocsvm_nu, max_iter=args.ocsvm_maxiter)
        self.seg_fuser = _OCSVM(random_state=42, nu=args.ocsvm_nu, max_iter=args.ocsvm_maxiter)
        self.s_lib, self.s_map_lib = [], []
        self.img_name = []
        self.save_num = 0
        self._engine = eng.Engine(self.deep_feature_extractor._vit, self.deep_feature_extractor.xyz_backbone.packed,
                                  size=self.xyz_size)
        self._banks = {}

    # ------------------------------------------------------------------------------------ extraction
    def __call__(self, rgb=None, xyz=None, out_type="rgb+xyz"):
        """features.py:123-158.  rgb [B,3,S,S], xyz [B,3,N] (unorganised, zeros removed).  Returns the
        reference's tuples; feature maps are copied to the CPU as the reference does, and each carries
        the device-resident Extraction in ``._cmdiad`` so the patch getters avoid a round trip."""
        want_rgb, want_xyz = "rgb" in out_type, "xyz" in out_type
        dev = self.device
        rgb_d = rgb.to(dev).float() if want_rgb else None
        if want_xyz:
            pts = xyz.to(dev).float().transpose(1, 2).contiguous()  # [B,N,3]
            # pixel indices are only known to the caller (nonzero_indices); get_xyz_patch builds pix2pt from them
            ex = eng.Extraction()
            ex.size = self.xyz_size
            with torch.no_grad():
                ex.rgb_tokens = self._engine.vit.forward_tokens(rgb_d) if want_rgb else None
                ex.xyz, ex.n_valid, ex.nz, ex.pix2pt = pts, None, None, None
                ex.xyz_feats, ex.center, ex.ori_idx, ex.center_idx = self._engine.pm.forward(pts)
                ex.idx3, ex.w3 = ops.interp3nn(pts, ex.center)
        else:
            with torch.no_grad():
                ex = self._engine.extract(rgb_d, want_xyz=False)
        out = []
        if want_rgb:
            B, T, C = ex.rgb_tokens.shape
            s = int(math.isqrt(T - 1))
            fmap = ex.rgb_tokens[:, 1:].permute(0, 2, 1).reshape(B, C, s, s).to("cpu")
            fmap._cmdiad = ex
            out.append([fmap])
        if want_xyz:
            xmap = ex.xyz_feats.transpose(1, 2).to("cpu")
            xmap._cmdiad = ex
            out += [[xmap], ex.center, ex.ori_idx, ex.center_idx, LazyInterpolated(ex)]
        return out[0] if len(out) == 1 else tuple(out)

    def _extract_device(self, rgb, organized_pc, want_rgb=True, want_xyz=True):
        """Device-resident form of (organized_pc_to_unorganized_pc_no_zeros + __call__) for the method classes' own use:
        the zero-pixel compaction runs in cmdiad_unorganize (bit-identical order to the host numpy form) and nothing is
        copied back to the host.  Returns the Extraction the patch getters consume."""
        dev = self.device
        ex = eng.Extraction()
        ex.size = self.xyz_size
        both = want_rgb and want_xyz and os.environ.get("CMDIAD_B1_OVERLAP", "1") != "0"
        with torch.no_grad():
            if want_xyz:
                opc = organized_pc.to(dev, torch.float32).contiguous()
                xyz, nz, pix2pt, nv = ops.unorganize(opc, None)
            if want_rgb and both:
                # the ViT goes to a second HIP stream: at B = 1 farthest-point sampling is a 2.7 ms chain on ONE compute unit
                # and the ViT (1.9 ms of small GEMMs) fits beside it.  Its ~100 launches are queued before the host waits
                # for the point count, so that wait costs nothing either.
                side = self.__dict__.get("_vit_stream")
                if side is None:
                    side = self.__dict__["_vit_stream"] = ops.shared_stream(dev, "features.vit")
                cur = torch.cuda.current_stream()
                rgb_dev = rgb.to(dev).float()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ex.rgb_tokens = self._engine.vit.forward_tokens(rgb_dev)
                rgb_dev.record_stream(side)
            if want_xyz:
                n = int(nv[0].item())  # the only host round trip: sizes the exact-N tensors the B = 1 path works on
                if n < self.args.group_size:
                    raise ValueError(f"point cloud has {n} valid points; the {self.args.group_size}-nearest-neighbour grouping "
                                     f"(models/models.py:88-113) needs at least {self.args.group_size}")
                ex.xyz, ex.nz, ex.pix2pt, ex.n_valid = xyz[:, :n].contiguous(), nz[:, :n], pix2pt, None
            if want_rgb and not both:
                ex.rgb_tokens = self._engine.vit.forward_tokens(rgb.to(dev).float())
            if want_xyz:
                ex.xyz_feats, ex.center, ex.ori_idx, ex.center_idx = self._engine.pm.forward(ex.xyz)
                ex.idx3, ex.w3 = ops.interp3nn(ex.xyz, ex.center)
            if want_rgb and both:
                cur.wait_stream(side)
                ex.rgb_tokens.record_stream(cur)
        return ex

    def get_rgb_patch(self, rgb_feature_maps):
        """features.py:160-167 -> (rgb_patch [784,768], rgb_patch2 [3136,768]) on the GPU."""
        ex = getattr(rgb_feature_maps[0], "_cmdiad", None)
        if ex is not None and len(rgb_feature_maps) == 1:
            return eng.Engine.rgb_patch(ex)[0], eng.Engine.rgb_patch56(ex)[0]
        fm = torch.cat(rgb_feature_maps, 1).to(self.device)
        p = fm.reshape(fm.shape[1], -1).T.contiguous()
        s = int(math.isqrt(p.shape[0]))
        p2 = p.reshape(s, 1, s, 1, -1).expand(s, 2, s, 2, -1).reshape(4 * s * s, -1)
        return p, p2

    def get_xyz_patch(self, xyz_feature_maps, interpolated_pc, nonzero_indices, get_2828=False):
        """features.py:169-184 -> [3136,768] (or [784,768] with get_2828) on the GPU."""
        if not isinstance(interpolated_pc, LazyInterpolated):
            raise TypeError("get_xyz_patch expects the LazyInterpolated handle returned by Features.__call__")
        ex = interpolated_pc.ex
        if ex.pix2pt is None:
            n = ex.xyz.shape[1]
            nz = torch.as_tensor(np.asarray(nonzero_indices), device=ex.xyz.device).long().view(1, n)
            ex.pix2pt = torch.full((1, self.xyz_size * self.xyz_size), -1, dtype=torch.int32, device=ex.xyz.device)
            ex.pix2pt.scatter_(1, nz, torch.arange(n, dtype=torch.int32, device=ex.xyz.device).view(1, n))
        return self._engine.xyz_patch(ex, P=28 if get_2828 else 56)[0]

    # ------------------------------------------------------------------------------------ scoring
    def calculate_dist(self, single_patch, patch_lib):
        assert len(single_patch.shape) == 2 and len(patch_lib.shape) == 2
        if self.args.dist_method_s != "l2":
            raise NotImplementedError("only dist_method_s='l2' is implemented (the reference's l1/cos branches "
                                      "pass CPU tensors to cupy and cannot run; SURVEY 2.2)")
        return DistHandle(single_patch, patch_lib)

    def _bank(self, lib):
        """Device bank (bf16 copy + squared norms) cached per library tensor."""
        key = (lib.data_ptr(), tuple(lib.shape), lib._version)
        hit = self._banks.get(lib.data_ptr())
        if hit is None or hit[0] != key:
            hit = (key, eng.Bank(lib.to(self.device).float()))
            self._banks[lib.data_ptr()] = hit
        return hit[1]

    def compute_single_s_s_map(self, patch, dist, feature_map_dims, modal='xyz'):
        """features.py:225-297 -> (s scalar tensor, s_map [1,gt,gt]) on the CPU (the blur is host PIL)."""
        lib = {"xyz": self.patch_xyz_lib, "rgb": self.patch_rgb_lib, "fusion": self.patch_fusion_lib,
               "share": self.patch_share_lib, "non_share": self.patch_non_share_lib}[modal]
        if isinstance(dist, DistHandle):
            # the nearest neighbours are searched in the modal library (as the reference's own call sites pair them); a
            # handle onto ANOTHER library would silently give scores that do not belong to `dist`
            if dist.lib is not lib and not (dist.lib.shape == lib.shape and dist.lib.data_ptr() == lib.data_ptr()):
                raise ValueError(f"compute_single_s_s_map(modal={modal!r}): `dist` was computed against a different library")
        elif dist is not None:
            raise TypeError("compute_single_s_s_map expects the DistHandle returned by calculate_dist (the Q x N matrix is "
                            "never built on this path); a materialised matrix cannot be re-used")
        r = eng.score_patches(patch.to(self.device).float().unsqueeze(0).contiguous(), self._bank(lib),
                              feature_map_dims, self.gt_size)
        s_map = self.blur(r["s_map_pre"].unsqueeze(0))  # [1,1,H,W] -> [1,H,W], 8-bit PIL blur (utils.py:71-83)
        return r["s"][0].cpu(), s_map

    def add_sample_to_mem_bank(self, sample):
        raise NotImplementedError

    def predict(self, sample, mask, label, rgb_path):
        raise NotImplementedError

    def interpolate_points(self, rgb, xyz):
        """features.py:216-219: the point-cloud feature maps and group centres of one forward pass, and the cloud itself."""
        rgb_feature_maps, xyz_feature_maps, center, ori_idx, center_idx, _ = self(rgb, xyz)
        return xyz_feature_maps, center, xyz

    def add_sample_to_late_fusion_mem_bank(self, sample):
        raise NotImplementedError

    def compute_s_s_map(self, *a, **k):
        raise NotImplementedError

    def run_coreset(self):
        raise NotImplementedError

    def calculate_metrics(self):
        """features.py:302-324."""
        from ..utils.au_pro_util import calculate_au_pro
        self.image_preds = np.stack(self.image_preds)
        self.image_labels = np.stack(self.image_labels)
        self.pixel_preds = np.array(self.pixel_preds)
        self.img_name = np.stack(self.img_name)
        if getattr(self.args, "save_raw_results", False):       # features.py:316-318 (the directory is created here)
            txt_to_save = np.concatenate((self.image_preds, self.image_labels, self.img_name), axis=1)
            os.makedirs(f'./visualization/{self.args.experiment_note}', exist_ok=True)
            np.savetxt(f'./visualization/{self.args.experiment_note}/{self.class_name}_raw_results.csv', txt_to_save, delimiter=',', fmt="%s")
        self.image_rocauc = roc_auc_score(self.image_labels, self.image_preds)
        self.pixel_rocauc = roc_auc_score(self.pixel_labels, self.pixel_preds)
        self.au_pro, _ = calculate_au_pro(self.gts, self.predictions)
        self.au_pro_001, _ = calculate_au_pro(self.gts, self.predictions, 0.01)

    def run_late_fusion(self):
        """features.py:352-358 (scikit-learn on the host by default; on the device with CMDIAD_OCSVM_DEVICE=1)."""
        self.s_lib = torch.cat(self.s_lib, 0)
        self.s_map_lib = torch.cat(self.s_map_lib, 0)
        self.detect_fuser.fit(self.s_lib)
        self.seg_fuser.fit(self.s_map_lib)

    def get_coreset_idx_randomp(self, z_lib, n=1000, eps=0.90, coreset_dtype='FP16', force_cpu=False, lib=''):
        """features.py:360-425: sparse random projection (host sklearn) + greedy k-centre selection.
        The greedy loop is the HIP kernel cmdiad_coreset_* (SURVEY 8f row f1)."""
        from .. import coreset
        print(f"   Fitting random projections. Start dim = {z_lib.shape}.")
        try:
            if os.environ.get("CMDIAD_PROJECT_HOST", "0") == "1":   # the reference's host transform (A/B runs, parity tests)
                transformer = random_projection.SparseRandomProjection(eps=eps, random_state=self.random_state)
                z = torch.tensor(transformer.fit_transform(z_lib.detach().cpu().numpy()))
            else:   # fitted by scikit-learn, transformed on the device: bit-identical (coreset.sparse_random_projection)
                z = coreset.sparse_random_projection(z_lib.detach().to(self.device), eps, self.random_state)
            print(f"   DONE.                 Transformed dim = {z.shape}.")
        except ValueError:
            print("   Error: could not project vectors. Please increase `eps`.")
            z = z_lib.detach()
        if self.args.dist_method_coreset != "l2":
            raise NotImplementedError("only dist_method_coreset='l2' is implemented")
        group = getattr(self, "coreset_group", None)
        if group is not None and coreset_dtype == "FP16":
            # row-sharded selection (fit-time sharding): every rank of the group must make this call with the same library
            return coreset.greedy_coreset_sharded(z.to(self.device), n, group).cpu()
        return coreset.greedy_coreset(z.to(self.device), n, coreset_dtype).cpu()
