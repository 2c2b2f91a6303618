"""`dropin_b1`: the B = 1 protocol the reference's main.py drives, through the drop-in classes."""
import os
import time

from .common import N_POINTS, RGB_ROWS, XYZ_ROWS

def dropin_b1(n=192, warm=32):
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
