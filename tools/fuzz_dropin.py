#!/usr/bin/env python3
"""Randomised EXACT invariant of the drop-in method classes (feature_extractors.multiple_features): whatever micro-batch size the
deferred calls are run at (CMDIAD_PREDICT_BATCH; 1 = the reference's strictly-per-call behaviour), the five-call protocol leaves the
same BITS behind -- libraries, statistics, late-fusion rows, image scores, pixel maps, metrics.  Random method class (RGB, Depth,
Point, DINO + Point-MAE, the two hallucination classes with either main modality), 2-5 training samples, 1-6 test samples,
clouds of a few thousand points ... no background, coreset on / off, the f_coreset greedy selection included.
    python tools/fuzz_dropin.py [seconds] [seed]     (synthetic weights; the oracle package only provides their state dicts)"""
import os
import sys
import time
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")
from cmdiad_amd.feature_extractors import multiple_features as mf  # noqa: E402
from cmdiad_amd.synth import synth_cloud, synth_rgb  # noqa: E402
from oracle import nets  # noqa: E402  (synthetic state dicts only)


def make_args(**kw):
    a = dict(rgb_backbone_name='vit_base_patch8_224_dino', xyz_backbone_name='Point_MAE', group_size=128, num_group=1024,
             rgb_size=224, xyz_size=224, gt_size=224, f_coreset=1.0, coreset_eps=0.9, coreset_dtype='FP16',
             random_state=0, dist_method_s='l2', dist_method_coreset='l2', main_modality='', use_hn=False,
             fusion_module_path='', ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0, xyz_smap_lambda=1.0,
             rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
             save_feature_for_fusion=False, save_seg_results=False, use_depth=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def run(cls, kw, train, tests, batch, sd):
    os.environ["CMDIAD_PREDICT_BATCH"] = str(batch)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = cls(make_args(**kw))
    m.deep_feature_extractor.rgb_backbone.load_state_dict(sd[0])
    m.deep_feature_extractor.xyz_backbone.load_state_dict(sd[1])
    if kw.get("use_hn"):
        m.fusion.load_state_dict(sd[2])
    for rgb, pc in train:
        m.add_sample_to_mem_bank((rgb, pc, rgb), class_name="synthetic")
    m.run_coreset()
    for rgb, pc in train:
        m.add_sample_to_late_fusion_mem_bank((rgb, pc, rgb))
    m.run_late_fusion()
    for k, (rgb, pc, mask) in enumerate(tests):
        m.predict((rgb, pc, rgb), mask, np.array([int(mask.any())]), [f"t{k}.png"])
    m.calculate_metrics()
    libs = [getattr(m, n) for n in ("patch_xyz_lib", "patch_rgb_lib", "patch_fusion_lib") if isinstance(getattr(m, n, None), torch.Tensor)]
    rows = lambda v: (v if isinstance(v, torch.Tensor) else torch.cat(list(v), 0)).detach().cpu()   # noqa: E731  (a tensor after run_late_fusion)
    return ([t.detach().cpu() for t in libs], rows(m.s_lib), rows(m.s_map_lib),
            np.concatenate([np.ravel(x) for x in m.image_preds]), np.stack(m.predictions), float(m.image_rocauc), float(m.pixel_rocauc),
            float(m.au_pro))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sd = (nets.synth_state_dict("vit", 31), nets.sharpen_pointmae(nets.synth_state_dict("pointmae", 21)), nets.synth_state_dict("halluc", 51))
    menu = [(mf.RGBFeatures, {}), (mf.DepthFeatures, {}), (mf.PointFeatures, {}), (mf.DoubleRGBPointFeatures, {}),
            (mf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="xyz")),
            (mf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="rgb"))]
    t0, n, seen = time.time(), 0, {}
    while time.time() - t0 < budget:
        cls, kw = menu[int(rs.randint(len(menu)))]
        kw = dict(kw)
        if rs.rand() < 0.4:
            kw.update(f_coreset=float(rs.choice([0.1, 0.5])))
        n_train, n_test = int(rs.randint(2, 6)), int(rs.randint(1, 7))

        def sample(i, anomalous):
            pc = synth_cloud(int(rs.randint(10 ** 6)), float(rs.choice([0.06, 0.2, 0.45, 0.7, 1.6])), texture=0.004)
            rgb = synth_rgb(int(rs.randint(10 ** 6)))
            mask = torch.zeros(1, 224, 224)
            if anomalous:
                pc[0, 2, 100:120, 100:120] -= 0.015 * (pc[0, 2, 100:120, 100:120] != 0)
                mask[0, 100:120, 100:120] = 1
            return rgb, pc, mask

        train = [sample(i, False)[:2] for i in range(n_train)]
        tests = [sample(i, i % 2 == 1) for i in range(n_test)]
        if not any(t[2].any() for t in tests) or all(t[2].any() for t in tests):
            tests.append(sample(99, not tests[0][2].any()))       # the metrics need both labels
        ref = run(cls, kw, train, tests, 1, sd)
        for batch in sorted({int(rs.choice([2, 3, 5])), 16}):
            got = run(cls, kw, train, tests, batch, sd)
            tag = (cls.__name__, kw, n_train, len(tests), batch)
            assert len(got[0]) == len(ref[0]) and all(torch.equal(a, b) for a, b in zip(got[0], ref[0])), ("libraries differ", tag)
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), ("late-fusion rows differ", tag)
            assert np.array_equal(got[3], ref[3]) and np.array_equal(got[4], ref[4]), ("predictions differ", tag)
            assert got[5:] == ref[5:], ("metrics differ", tag, got[5:], ref[5:])
        n += 1
        seen[cls.__name__] = seen.get(cls.__name__, 0) + 1
    print("dropin fuzz ok", n, seen, flush=True)


if __name__ == "__main__":
    main()
