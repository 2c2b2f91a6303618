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

import numpy as np
import torch

from .. import engine as eng
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


class _MethodBase(Features):
    """Shared plumbing of the method classes (not part of the reference's public surface)."""

    def _extract(self, sample, want_rgb=True, want_xyz=True):
        """(rgb_maps, xyz_maps, interpolated, nonzero_indices) as the reference's methods name them.  The two map lists
        are handles onto the device-resident extraction (the public Features.__call__ still returns the CPU copies the
        reference returns; the method classes themselves never need them)."""
        from .features import LazyInterpolated
        ex = self._extract_device(sample[0], sample[1], want_rgb, want_xyz)
        handle = torch.empty(0)
        handle._cmdiad = ex
        return [handle], [handle], LazyInterpolated(ex), None

    def _coreset(self, lib, name):
        if self.f_coreset < 1:
            idx = self.get_coreset_idx_randomp(lib, n=int(self.f_coreset * lib.shape[0]), eps=self.coreset_eps,
                                               lib=name, coreset_dtype=self.coreset_dtype)
            self.coreset_idx = idx
            return lib[idx.to(lib.device)]
        return lib

    def _score(self, patch, mean, std, modal):
        patch = eng.normalize(patch.to(self.device).float(), mean, std)  # a11: fused HIP kernel
        dist = self.calculate_dist(patch, getattr(self, f"patch_{modal}_lib"))
        return self.compute_single_s_s_map(patch, dist, _side(patch), modal=modal)

    def _score_many(self, items):
        """[(patch, mean, std, modal)] -> [(s, s_map)], each pair as _score returns it.  The device work of every modality
        is queued first, then ONE blur launch (a block per map) and ONE device->host copy: at B = 1 the per-modality
        blur + copy of compute_single_s_s_map left the GPU idle while the host queued the next modality."""
        from .. import ops
        gt = self.gt_size
        rs = []
        for patch, mean, std, modal in items:
            patch = eng.normalize(patch.to(self.device).float(), mean, std)
            rs.append(eng.score_patches(patch.unsqueeze(0).contiguous(), self._bank(getattr(self, f"patch_{modal}_lib")),
                                        _side(patch), gt))
        k = len(rs)
        maps = ops.blur8_maps(torch.cat([r["s_map_pre"].reshape(1, gt, gt) for r in rs]).contiguous(), float(self.blur.radius))
        host = torch.cat([maps.reshape(-1)] + [r["s"][:1].float() for r in rs]).cpu()
        return [(host[k * gt * gt + i], host[i * gt * gt:(i + 1) * gt * gt].view(1, gt, gt)) for i in range(k)]

    def _fuse_inputs(self, pairs):
        """pairs: [(lambda_s, s, lambda_map, s_map)] -> (s [1,k], s_map [gt*gt, k]) as the reference stacks them."""
        s = torch.tensor([[float(ls * sv) for ls, sv, _, _ in pairs]])
        # numpy for the two host-side products: torch's CPU intra-op pool (one thread per core by default) stalls for
        # 80 ms every few calls on 50 176-element tensors on a 128-core host; same float32 arithmetic either way
        s_map = torch.from_numpy(np.stack([(np.float32(lm) * m.numpy()).reshape(-1) for _, _, lm, m in pairs], axis=1))
        return s, s_map

    def _record(self, s, s_map, mask, label, rgb_path):
        # from_numpy, not torch.tensor(): no 50 176-element copy through torch's CPU thread pool (see _fuse_inputs)
        s = torch.from_numpy(np.ascontiguousarray(self.detect_fuser.score_samples(s.numpy())))
        s_map = torch.from_numpy(np.ascontiguousarray(self.seg_fuser.score_samples(s_map.numpy()))).view(1, self.gt_size, self.gt_size)
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
    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.class_name = class_name
        rgb_maps, _, _, _ = self._extract(sample, want_xyz=False)
        self.patch_rgb_lib.append(self.get_rgb_patch(rgb_maps)[0])

    def run_coreset(self):
        self.patch_rgb_lib = torch.cat(self.patch_rgb_lib, 0)
        self.rgb_mean, self.rgb_std = torch.mean(self.patch_rgb_lib), torch.std(self.patch_rgb_lib)
        self.patch_rgb_lib = self._coreset(eng.normalize(self.patch_rgb_lib, self.rgb_mean, self.rgb_std), 'patch_rgb_lib')

    def _s(self, sample):
        rgb_maps, _, _, _ = self._extract(sample, want_xyz=False)
        s_rgb, m_rgb = self._score(self.get_rgb_patch(rgb_maps)[0], self.rgb_mean, self.rgb_std, 'rgb')
        return self._fuse_inputs([(self.args.rgb_s_lambda, s_rgb, self.args.rgb_smap_lambda, m_rgb)])

    def add_sample_to_late_fusion_mem_bank(self, sample):
        s, s_map = self._s(sample)
        self.s_lib.append(s)
        self.s_map_lib.append(s_map)

    def predict(self, sample, mask, label, rgb_path):
        s, s_map = self._s(sample)
        self._record(s, s_map, mask, label, rgb_path)


class PointFeatures(_MethodBase):
    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.class_name = class_name
        _, xyz_maps, interp, nz = self._extract(sample, want_rgb=False)
        self.patch_xyz_lib.append(self.get_xyz_patch(xyz_maps, interp, nz))

    def run_coreset(self):
        self.patch_xyz_lib = torch.cat(self.patch_xyz_lib, 0)
        self.xyz_mean, self.xyz_std = torch.mean(self.patch_xyz_lib), torch.std(self.patch_xyz_lib)
        self.patch_xyz_lib = self._coreset(eng.normalize(self.patch_xyz_lib, self.xyz_mean, self.xyz_std), 'patch_xyz_lib')

    def _s(self, sample):
        _, xyz_maps, interp, nz = self._extract(sample, want_rgb=False)
        s_xyz, m_xyz = self._score(self.get_xyz_patch(xyz_maps, interp, nz), self.xyz_mean, self.xyz_std, 'xyz')
        return self._fuse_inputs([(self.args.xyz_s_lambda, s_xyz, self.args.xyz_smap_lambda, m_xyz)])

    def add_sample_to_late_fusion_mem_bank(self, sample):
        s, s_map = self._s(sample)
        self.s_lib.append(s)
        self.s_map_lib.append(s_map)

    def predict(self, sample, mask, label, rgb_path):
        s, s_map = self._s(sample)
        self._record(s, s_map, mask, label, rgb_path)


class DoubleRGBPointFeatures(_MethodBase):
    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.class_name = class_name
        rgb_maps, xyz_maps, interp, nz = self._extract(sample)
        xyz_patch = self.get_xyz_patch(xyz_maps, interp, nz)
        rgb_patch, rgb_patch2 = self.get_rgb_patch(rgb_maps)
        if getattr(self.args, "save_feature_for_fusion", False):
            # the trainer's on-disk format: [3136, 768 xyz | 768 rgb] f32 per sample (multiple_features.py:815-825)
            for sub in ("", "train", "test"):
                os.makedirs(os.path.join(self.args.save_path, sub), exist_ok=True)
            torch.save(torch.cat([xyz_patch, rgb_patch2], dim=1).cpu(),
                       os.path.join(self.args.save_path, 'train', class_name + str(self.ins_id) + '.pt'))
            self.ins_id += 1
        self.patch_xyz_lib.append(xyz_patch)
        self.patch_rgb_lib.append(rgb_patch)

    def run_coreset(self):
        self.patch_xyz_lib = torch.cat(self.patch_xyz_lib, 0)
        self.patch_rgb_lib = torch.cat(self.patch_rgb_lib, 0)
        # cross-wired exactly as the reference (multiple_features.py:877-880)
        self.xyz_mean = torch.mean(self.patch_xyz_lib)
        self.xyz_std = torch.std(self.patch_rgb_lib)
        self.rgb_mean = torch.mean(self.patch_xyz_lib)
        self.rgb_std = torch.std(self.patch_rgb_lib)
        self.patch_xyz_lib = self._coreset(eng.normalize(self.patch_xyz_lib, self.xyz_mean, self.xyz_std), 'patch_xyz_lib')
        self.patch_rgb_lib = self._coreset(eng.normalize(self.patch_rgb_lib, self.rgb_mean, self.rgb_std), 'patch_rgb_lib')

    def _s(self, sample, test=False):
        if getattr(self.args, "use_depth", False):
            sample[0] = sample[1]
        rgb_maps, xyz_maps, interp, nz = self._extract(sample)
        xyz_patch = self.get_xyz_patch(xyz_maps, interp, nz)
        rgb_patch, rgb_patch2 = self.get_rgb_patch(rgb_maps)
        if test and getattr(self.args, "save_feature_for_fusion", False):
            torch.save(torch.cat([xyz_patch, rgb_patch2], dim=1).cpu(),
                       os.path.join(self.args.save_path, 'test', self.class_name + str(self.ins_id) + '.pt'))
            self.ins_id += 1
        (s_xyz, m_xyz), (s_rgb, m_rgb) = self._score_many([(xyz_patch, self.xyz_mean, self.xyz_std, 'xyz'),
                                                            (rgb_patch, self.rgb_mean, self.rgb_std, 'rgb')])
        return self._fuse_inputs([(self.args.xyz_s_lambda, s_xyz, self.args.xyz_smap_lambda, m_xyz),
                                  (self.args.rgb_s_lambda, s_rgb, self.args.rgb_smap_lambda, m_rgb)])

    def add_sample_to_late_fusion_mem_bank(self, sample):
        s, s_map = self._s(sample)
        self.s_lib.append(s)
        self.s_map_lib.append(s_map)

    def predict(self, sample, mask, label, rgb_path):
        s, s_map = self._s(sample, test=True)
        self._record(s, s_map, mask, label, rgb_path)


class RGBorXYZWithOneHallucination(_MethodBase):
    """MTFI with one real and one hallucinated modality (multiple_features.py:312-573): the main modality's real features
    plus the other modality's features hallucinated either from the main modality's FEATURES (``--use_hn``: the FtoF MLP, or
    the FtoF conv head when ``--use_hn_conv`` is given as well) or from the main modality's INPUT (``--use_hrnet``, ItoF)."""

    def _hallucinate(self, sample, xyz_patch, rgb_patch2):
        a = self.args
        if a.main_modality not in ('rgb', 'xyz'):
            raise Exception('Unknown modality')
        with torch.no_grad():
            if getattr(a, "use_hrnet", False):  # multiple_features.py:326-331, 343-348: from the raw image / point map
                src = sample[0] if a.main_modality == 'rgb' else sample[1]
                h = self.fusion.hallucination_tokens(src.to(self.device))
                assert tuple(h.shape[1:]) == (3136, 768)
            elif a.main_modality == 'rgb':
                h = self.fusion.hallucination_generation(rgb_feature=rgb_patch2.unsqueeze(0), out_type='xyz')
            else:
                h = self.fusion.hallucination_generation(xyz_feature=xyz_patch.unsqueeze(0), out_type='rgb')
        assert len(h.shape) == 3
        return h.reshape(-1, h.shape[2]).detach()

    def _patches(self, sample, fit=False):
        rgb_maps, xyz_maps, interp, nz = self._extract(sample)
        xyz_patch = self.get_xyz_patch(xyz_maps, interp, nz)
        rgb_patch, rgb_patch2 = self.get_rgb_patch(rgb_maps)
        return xyz_patch, rgb_patch, self._hallucinate(sample, xyz_patch, rgb_patch2)

    def add_sample_to_mem_bank(self, sample, class_name=None):
        self.class_name = class_name
        xyz_patch, rgb_patch, hall = self._patches(sample, fit=True)
        self.patch_rgb_lib.append(rgb_patch)
        self.patch_xyz_lib.append(xyz_patch)
        self.patch_fusion_lib.append(hall)

    def run_coreset(self):
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

    def _s(self, sample):
        xyz_patch, rgb_patch, hall = self._patches(sample)
        a = self.args
        main_item = ((rgb_patch, self.rgb_mean, self.rgb_std, 'rgb') if a.main_modality == 'rgb'
                     else (xyz_patch, self.xyz_mean, self.xyz_std, 'xyz'))
        (s_f, m_f), (s_m, m_m) = self._score_many([(hall, self.fusion_mean, self.fusion_std, 'fusion'), main_item])
        if a.main_modality == 'rgb':
            main = (a.rgb_s_lambda, s_m, a.rgb_smap_lambda, m_m)
        else:
            main = (a.xyz_s_lambda, s_m, a.xyz_smap_lambda, m_m)
        return self._fuse_inputs([main, (a.fusion_s_lambda, s_f, a.fusion_smap_lambda, m_f)])

    def add_sample_to_late_fusion_mem_bank(self, sample):
        s, s_map = self._s(sample)
        self.s_lib.append(s)
        self.s_map_lib.append(s_map)

    def predict(self, sample, mask, label, rgb_path):
        s, s_map = self._s(sample)
        self._record(s, s_map, mask, label, rgb_path)


class RGBorXYZWithOneHallucinationFromFeature(RGBorXYZWithOneHallucination):
    """MTFI feature-to-INPUT (multiple_features.py:576-797; ``--use_hn_from_rgb_mlp`` / ``--use_hn_from_rgb_conv``): the head
    turns the main modality's features into the OTHER modality's input -- an organised point map [1,3,224,224] from rgb
    features, or an RGB image from xyz features -- and the frozen extractor of that modality is run on it; those
    re-extracted features are the "hallucination" library / query.  Banks, statistics, coreset and scoring are
    the parent's (the reference repeats them verbatim, :614-648, :754-797).

    As in the reference, with main_modality == 'rgb' the real point cloud is only read while the memory bank is built
    (its patches feed the cross-wired statistics, :582,605,612-618); late fusion and predict never touch it (:651-663,
    :701-719), so the 3-D branch of the extractor is skipped there."""

    def _patches(self, sample, fit=False):
        a = self.args
        if a.main_modality == 'rgb':
            ex = self._extract_device(sample[0], sample[1], want_rgb=True, want_xyz=fit)
            rgb_patch, rgb_patch2 = eng.Engine.rgb_patch(ex)[0], eng.Engine.rgb_patch56(ex)[0]
            xyz_patch = self._engine.xyz_patch(ex, P=56)[0] if fit else None
            with torch.no_grad():
                pc = self.fusion.hallucination_generation(rgb_patch2.unsqueeze(0))  # [1,3,224,224] hallucinated point map
            assert tuple(pc.shape) == (1, 3, self.xyz_size, self.xyz_size), tuple(pc.shape)
            hx = self._extract_device(None, pc, want_rgb=False, want_xyz=True)   # zero-coordinate pixels dropped as :592-594
            hall = self._engine.xyz_patch(hx, P=56)[0]
        elif a.main_modality == 'xyz':
            ex = self._extract_device(sample[0], sample[1], want_rgb=fit, want_xyz=True)
            xyz_patch = self._engine.xyz_patch(ex, P=56)[0]
            rgb_patch = eng.Engine.rgb_patch(ex)[0] if fit else None
            with torch.no_grad():
                img = self.fusion.hallucination_generation(xyz_patch.unsqueeze(0))
            assert tuple(img.shape) == tuple(sample[0].shape), (tuple(img.shape), tuple(sample[0].shape))
            hall = eng.Engine.rgb_patch(self._extract_device(img, None, want_rgb=True, want_xyz=False))[0]
        else:
            raise NotImplementedError
        return xyz_patch, rgb_patch, hall
