#!/usr/bin/env python3
"""Throughput of the B = 1 drop-in protocol (the way cmdiad_runner.py drives the reference): DoubleRGBPointFeatures.predict
per image with bagel-sized libraries, host-side inputs (so H2D of the sample and D2H of the maps are included).
Secondary number next to bench.py (which measures the batched engine, BASELINE configs[1])."""
import os
import sys
import time
import types
import warnings

import torch

os.environ.setdefault("CMDIAD_ALLOW_RANDOM_INIT", "1")  # synthetic weights: no checkpoints offline

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures  # noqa: E402
from cmdiad_amd.synth import synth_bank, synth_cloud_fixed_n, synth_rgb  # noqa: E402


def main(n=40, warm=5):
    a = dict(rgb_backbone_name='vit_base_patch8_224_dino', xyz_backbone_name='Point_MAE', group_size=128, num_group=1024,
             rgb_size=224, xyz_size=224, gt_size=224, f_coreset=1.0, coreset_eps=0.9, coreset_dtype='FP16',
             random_state=None, dist_method_s='l2', dist_method_coreset='l2', main_modality='', use_hn=False,
             fusion_module_path='', ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0, xyz_smap_lambda=1.0,
             rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
             save_feature_for_fusion=False, save_seg_results=False, use_depth=False)
    from cmdiad_amd.utils.utils import set_multithreading
    set_multithreading(6)  # main.py:149,190-191: the reference's default --cpu_core_num
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DoubleRGBPointFeatures(types.SimpleNamespace(**a))
    m.patch_xyz_lib = synth_bank(76518, 768, 4321).cuda()
    m.patch_rgb_lib = synth_bank(19129, 768, 4322).cuda()
    m.xyz_mean = m.rgb_mean = torch.tensor(0.0)
    m.xyz_std = m.rgb_std = torch.tensor(1.0)
    from sklearn import linear_model
    import numpy as np
    rs = np.random.RandomState(0)
    m.detect_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 2))
    m.seg_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 2))
    samples = [(synth_rgb(i), synth_cloud_fixed_n(1000 + i, 24576)) for i in range(8)]
    mask = torch.zeros(1, 224, 224)
    for i in range(warm):
        rgb, pc = samples[i % 8]
        m.predict((rgb, pc, pc), mask, 0, ["x.png"])
    m._flush("predict")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        rgb, pc = samples[i % 8]
        m.predict((rgb, pc, pc), mask, 0, ["x.png"])
    m._flush("predict")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{{"dropin_B1_images_per_s": {n / dt:.2f}, "ms_per_image": {dt / n * 1e3:.2f}}}')


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:3]))      # [images] [warm-up images]; CMDIAD_PREDICT_BATCH selects the micro-batch size
