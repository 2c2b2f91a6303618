"""CPU end-to-end path composed the way the reference composes it (TEST INFRASTRUCTURE ONLY; also the
``cpu_baseline`` of bench.py, kind "port"): torch-CPU for the networks, ``torch.cdist``, pooling and
scoring exactly as feature_extractors/features.py and multiple_features.py call them, the C oracle
for FPS / kNN grouping (the reference has no CPU implementation of those: models/models.py:5-6
hard-imports the CUDA packages).

Protocols mirrored: DoubleRGBPointFeatures (multiple_features.py:800-1015) = "DINO+Point_MAE"; RGBFeatures (:28-121) and
PointFeatures (:207-309), the single-library classes with their OWN (correctly wired) statistics; and
RGBorXYZWithOneHallucination (:312-573), the MTFI feature-to-feature class (main modality + hallucinated other modality,
statistics cross-wired as SURVEY F5).  Pinned by tests/golden/g6_protocol.npz and g11_methods.npz (the reference's own
classes driven over the same backbone restatements).
"""
import math
import time

import numpy as np
import torch

from . import kernels as ok
from . import nets, scoring


class CpuExtractor:
    def __init__(self, sd_vit, sd_pm, num_group=1024, group_size=128):
        self.sd_vit, self.sd_pm, self.G, self.M = sd_vit, sd_pm, num_group, group_size
        self.timing = {}

    def _t(self, key, t0):
        self.timing[key] = self.timing.get(key, 0.0) + time.perf_counter() - t0

    def __call__(self, rgb, organized_pc):
        """rgb [1,3,S,S], organized_pc [1,3,S,S] -> (rgb_patch [784,768], xyz_patch [3136,768])."""
        with torch.no_grad():
            t0 = time.perf_counter()
            pc, nz = scoring.unorganize_no_zeros(organized_pc)                    # a1
            self._t("unorganize", t0); t0 = time.perf_counter()
            fmap = nets.vit_forward(self.sd_vit, rgb)                              # a2
            self._t("vit", t0); t0 = time.perf_counter()
            xyz = np.ascontiguousarray(pc[0].T.numpy())[None]
            cidx, cen = ok.fps(xyz, self.G)                                        # a3
            self._t("fps", t0); t0 = time.perf_counter()
            _, nb = ok.knn_group(xyz, cen, self.M)                                 # a4
            self._t("knn_group", t0); t0 = time.perf_counter()
            tok = nets.pointmae_encoder(self.sd_pm, torch.from_numpy(nb))          # a5
            self._t("encoder", t0); t0 = time.perf_counter()
            center = torch.from_numpy(cen)
            feats = nets.pointmae_transformer(self.sd_pm, tok, center)             # a6  [1,768,G]
            self._t("pmae_transformer", t0); t0 = time.perf_counter()
            interp = scoring.interpolating_points(pc, center.permute(0, 2, 1), feats)  # a7
            self._t("interp", t0); t0 = time.perf_counter()
            xyz_patch = scoring.get_xyz_patch(interp, nz)                          # a9
            self._t("xyz_pool", t0); t0 = time.perf_counter()
            rgb_patch, _ = scoring.get_rgb_patch(fmap)                             # a10
            self._t("rgb_patch", t0)
        return rgb_patch.contiguous(), xyz_patch.contiguous()


class CpuDoubleRGBPoint:
    """fit (banks + cross-wired statistics, optional greedy coreset) and predict (pre-OCSVM scores)."""

    def __init__(self, extractor, lambdas=(1.0, 1.0, 0.1, 0.1), f_coreset=1.0, coreset_eps=0.9, random_state=None):
        self.ex = extractor
        self.f_coreset, self.coreset_eps, self.random_state = f_coreset, coreset_eps, random_state
        self.xyz_s_l, self.xyz_m_l, self.rgb_s_l, self.rgb_m_l = lambdas
        self.timing = {}

    def fit(self, samples, coreset_override=None):
        """coreset_override = (xyz_idx, rgb_idx): use these selections instead of running the greedy coreset (the
        selection is chaotic in the last ulp of its input; tests pin it separately, g9_coreset)."""
        rp, xp = zip(*[self.ex(r, p) for r, p in samples])
        xyz_lib, rgb_lib = torch.cat(xp, 0), torch.cat(rp, 0)
        # multiple_features.py:877-880 (cross-wired)
        self.xyz_mean = self.rgb_mean = torch.mean(xyz_lib)
        self.xyz_std = self.rgb_std = torch.std(rgb_lib)
        self.xyz_lib = (xyz_lib - self.xyz_mean) / self.xyz_std
        self.rgb_lib = (rgb_lib - self.rgb_mean) / self.rgb_std
        if self.f_coreset < 1:  # multiple_features.py:885-895
            self.coreset_idx = {}
            for k, name in enumerate(("xyz_lib", "rgb_lib")):
                lib = getattr(self, name)
                idx = scoring.coreset_idx_randomp(lib, int(self.f_coreset * lib.shape[0]), self.coreset_eps, self.random_state)
                self.coreset_idx[name] = idx
                setattr(self, name, lib[idx if coreset_override is None else torch.as_tensor(coreset_override[k]).long()])
        return list(zip(rp, xp))

    def set_banks(self, xyz_lib, rgb_lib, xyz_mean, xyz_std, rgb_mean, rgb_std):
        self.xyz_lib, self.rgb_lib = xyz_lib, rgb_lib
        self.xyz_mean, self.xyz_std, self.rgb_mean, self.rgb_std = xyz_mean, xyz_std, rgb_mean, rgb_std

    def score(self, rgb_patch, xyz_patch, blur=True):
        t0 = time.perf_counter()
        rx = scoring.score_modality(xyz_patch, self.xyz_lib, self.xyz_mean, self.xyz_std, blur=blur)
        self.timing["score_xyz"] = self.timing.get("score_xyz", 0.0) + time.perf_counter() - t0
        t0 = time.perf_counter()
        rr = scoring.score_modality(rgb_patch, self.rgb_lib, self.rgb_mean, self.rgb_std, blur=blur)
        self.timing["score_rgb"] = self.timing.get("score_rgb", 0.0) + time.perf_counter() - t0
        s = torch.tensor([[self.xyz_s_l * rx["s"], self.rgb_s_l * rr["s"]]])
        s_map = torch.cat([self.xyz_m_l * rx["s_map"], self.rgb_m_l * rr["s_map"]], 0).reshape(2, -1).permute(1, 0)
        return s, s_map, rx, rr

    def predict(self, rgb, organized_pc, blur=True):
        rp, xp = self.ex(rgb, organized_pc)
        return self.score(rp, xp, blur=blur)


def _coreset(lib, f_coreset, eps, random_state, override=None):
    """features.py:360-425 as the method classes call it (`n = int(f_coreset * rows)`); returns (rows kept, own selection)."""
    if f_coreset >= 1:
        return lib, None
    idx = scoring.coreset_idx_randomp(lib, int(f_coreset * lib.shape[0]), eps, random_state)
    return lib[idx if override is None else torch.as_tensor(override).long()], idx


class CpuSingleModality:
    """RGBFeatures (multiple_features.py:28-121) / PointFeatures (:207-309): one library, statistics from that library
    (:38-41, :221-223), one (s, s_map) column."""

    def __init__(self, extractor, modality, lambdas=(1.0, 1.0), f_coreset=1.0, coreset_eps=0.9, random_state=None):
        assert modality in ("rgb", "xyz")
        self.ex, self.modality = extractor, modality
        self.s_l, self.m_l = lambdas
        self.f_coreset, self.coreset_eps, self.random_state = f_coreset, coreset_eps, random_state

    def _patch(self, rgb, organized_pc):
        rp, xp = self.ex(rgb, organized_pc)
        return rp if self.modality == "rgb" else xp

    def fit(self, samples, coreset_override=None):
        feats = [self._patch(r, p) for r, p in samples]
        lib = torch.cat(feats, 0)
        self.mean, self.std = torch.mean(lib), torch.std(lib)
        self.lib, self.coreset_idx = _coreset((lib - self.mean) / self.std, self.f_coreset, self.coreset_eps, self.random_state,
                                              coreset_override)
        return feats

    def score(self, patch, blur=True):
        r = scoring.score_modality(patch, self.lib, self.mean, self.std, blur=blur)
        s = torch.tensor([[self.s_l * r["s"]]])
        s_map = (self.m_l * r["s_map"]).reshape(1, -1).permute(1, 0)
        return s, s_map, r

    def predict(self, rgb, organized_pc, blur=True):
        return self.score(self._patch(rgb, organized_pc), blur=blur)


class CpuOneHallucination:
    """RGBorXYZWithOneHallucination with --use_hn (multiple_features.py:312-573): the main modality's real patches plus the
    OTHER modality's features hallucinated from them by the distilled MLP (hallucination_network.py:34-45); libraries
    `main` and `fusion`; every mean from the xyz library and every std from the rgb library (:372-377);
    s = [main, fusion] columns (:459-470)."""

    def __init__(self, extractor, sd_halluc, main_modality, lambdas=(1.0, 1.0, 1.0, 1.0), f_coreset=1.0, coreset_eps=0.9,
                 random_state=None):
        assert main_modality in ("rgb", "xyz")
        self.ex, self.sd_h, self.main = extractor, sd_halluc, main_modality
        self.main_s_l, self.main_m_l, self.fus_s_l, self.fus_m_l = lambdas
        self.f_coreset, self.coreset_eps, self.random_state = f_coreset, coreset_eps, random_state

    def patches(self, rgb, organized_pc):
        """-> (main patch, hallucinated patch [3136,768]).  main rgb: rgb_patch [784,768] scored on 28 x 28, the
        hallucination is generated from rgb_patch2 (56 x 56 replication, :345); main xyz: from the xyz patch (:360)."""
        rp, xp = self.ex(rgb, organized_pc)
        with torch.no_grad():
            if self.main == "rgb":
                C = rp.shape[1]
                _, rp2 = scoring.get_rgb_patch(rp.T.reshape(1, C, 28, 28))
                hall = nets.halluc_generate(self.sd_h, rp2.unsqueeze(0), "rgb2xyz")
                return rp, xp, hall[0]
            hall = nets.halluc_generate(self.sd_h, xp.unsqueeze(0), "xyz2rgb")
            return rp, xp, hall[0]

    def fit(self, samples, coreset_override=None):
        trip = [self.patches(r, p) for r, p in samples]
        rgb_lib, xyz_lib, fus_lib = (torch.cat([t[k] for t in trip], 0) for k in range(3))
        self.mean, self.std = torch.mean(xyz_lib), torch.std(rgb_lib)      # :372-377, all three pairs
        main_lib = rgb_lib if self.main == "rgb" else xyz_lib
        ov = coreset_override or (None, None)
        self.main_lib, self.main_coreset = _coreset((main_lib - self.mean) / self.std, self.f_coreset, self.coreset_eps,
                                                    self.random_state, ov[0])
        self.fus_lib, self.fus_coreset = _coreset((fus_lib - self.mean) / self.std, self.f_coreset, self.coreset_eps,
                                                  self.random_state, ov[1])
        return trip

    def score(self, rp, xp, hall, blur=True):
        rf = scoring.score_modality(hall, self.fus_lib, self.mean, self.std, blur=blur)
        rm = scoring.score_modality(rp if self.main == "rgb" else xp, self.main_lib, self.mean, self.std, blur=blur)
        s = torch.tensor([[self.main_s_l * rm["s"], self.fus_s_l * rf["s"]]])
        s_map = torch.cat([self.main_m_l * rm["s_map"], self.fus_m_l * rf["s_map"]], 0).reshape(2, -1).permute(1, 0)
        return s, s_map, rm, rf

    def predict(self, rgb, organized_pc, blur=True):
        return self.score(*self.patches(rgb, organized_pc), blur=blur)
