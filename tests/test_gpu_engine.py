"""GPU: the drop-in method classes and the batched engine end to end against the CPU oracle pipeline
(oracle/pipeline.py) on a tiny synthetic class, plus batch-size invariance and the sharded-bank merge."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import engine as eng  # noqa: E402
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.synth import synth_cloud, synth_rgb  # noqa: E402
from oracle import nets  # noqa: E402
from oracle.pipeline import CpuDoubleRGBPoint, CpuExtractor  # noqa: E402

DEV = "cuda"


def make_args(**kw):
    a = dict(rgb_backbone_name='vit_base_patch8_224_dino', xyz_backbone_name='Point_MAE', group_size=128, num_group=1024,
             rgb_size=224, xyz_size=224, gt_size=224, f_coreset=1.0, coreset_eps=0.9, coreset_dtype='FP16',
             random_state=None, dist_method_s='l2', dist_method_coreset='l2', main_modality='', use_hn=False,
             fusion_module_path='', ocsvm_nu=0.5, ocsvm_maxiter=1000, xyz_s_lambda=1.0, xyz_smap_lambda=1.0,
             rgb_s_lambda=0.1, rgb_smap_lambda=0.1, fusion_s_lambda=1.0, fusion_smap_lambda=1.0,
             save_feature_for_fusion=False, save_seg_results=False, use_depth=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def synth_sample(i, anomalous=False):
    pc = synth_cloud(200 + i, 0.30 + 0.02 * (i % 3), texture=0.004)
    rgb = synth_rgb(i)
    if anomalous:
        pc[0, 2, 100:120, 100:120] -= 0.005 * (pc[0, 2, 100:120, 100:120] != 0)  # 5 mm dent
        rgb[0, :, 100:120, 100:120] += 2.0
    return rgb, pc


import functools  # noqa: E402

from conftest import prefetched  # noqa: E402


@functools.lru_cache(maxsize=None)
def _weights():
    return nets.synth_state_dict("vit", 31), nets.synth_state_dict("pointmae", 21)


@pytest.fixture(scope="module")
def weights():
    return _weights()


def oracle_fit():
    """The CPU oracle's fit on the 4 train samples (no GPU call); started in the background at collection (conftest.prefetched)."""
    sd_vit, sd_pm = _weights()
    train = [synth_sample(i) for i in range(4)]
    cpu = CpuDoubleRGBPoint(CpuExtractor(sd_vit, sd_pm))
    cpu_feats = cpu.fit(train)
    return cpu, cpu_feats, train


@pytest.fixture(scope="module")
def fitted(weights):
    """CPU oracle fit + the drop-in's fit on the same 4 train samples."""
    import warnings
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures
    sd_vit, sd_pm = weights
    cpu, cpu_feats, train = prefetched(oracle_fit)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DoubleRGBPointFeatures(make_args())
    m.deep_feature_extractor.rgb_backbone.load_state_dict(sd_vit)
    m.deep_feature_extractor.xyz_backbone.load_state_dict(sd_pm)
    for rgb, pc in train:
        m.add_sample_to_mem_bank((rgb, pc, pc), class_name="synth")
    m.run_coreset()
    return cpu, cpu_feats, m, train


def _rel(got, ref):
    scale = ref.abs().mean().item()
    err = (got - ref).abs()
    return err.mean().item() / scale, err.max().item() / scale


@pytest.mark.oracle_prefetch("oracle_fit")
def test_fit_banks_and_statistics(fitted):
    cpu, cpu_feats, m, _ = fitted
    # F5 cross-wired scalar statistics
    assert abs(float(m.xyz_mean) - float(cpu.xyz_mean)) < 5e-3 * abs(float(cpu.xyz_mean)) + 1e-4
    assert abs(float(m.xyz_std) - float(cpu.xyz_std)) < 5e-3 * float(cpu.xyz_std)
    assert float(m.rgb_mean) == float(m.xyz_mean) and float(m.rgb_std) == float(m.xyz_std)
    assert m.patch_xyz_lib.shape == (4 * 3136, 768) and m.patch_rgb_lib.shape == (4 * 784, 768)
    # patch features: bf16 networks vs fp32 oracle (tolerance model of tests/test_gpu_nets.py)
    mr, xr = _rel(m.patch_rgb_lib.cpu(), cpu.rgb_lib)
    assert mr < 0.02 and xr < 0.2, (mr, xr)
    mx, xx = _rel(m.patch_xyz_lib.cpu(), cpu.xyz_lib)
    assert mx < 0.02 and xx < 0.2, (mx, xx)


def test_predict_scores_match_oracle(fitted):
    cpu, _, m, _ = fitted
    for i, anomalous in ((10, False), (11, True)):
        rgb, pc = synth_sample(i, anomalous)
        s_ref, map_ref, rx, rr = cpu.predict(rgb, pc)
        s_got, map_got = m._score_batch([(rgb, pc, pc)])[0]
        # image-level pre-OCSVM scores: w * s_star of each modality (2-3 % from the bf16 feature error)
        # (with random-init weights on a smooth synthetic surface the xyz features of neighbouring patches
        # are nearly identical, so s_xyz = w * s_star is a small difference of near-equal numbers: absolute
        # tolerance at 5 % of the normalised feature scale; the scorer itself is pinned exactly below)
        np.testing.assert_allclose(s_got.numpy(), s_ref.numpy(), rtol=4e-2, atol=0.05)
        # 224x224 blurred maps (8-bit quantised): compare at 2 grey levels of the map maximum
        for col in range(2):
            a, b = map_got[:, col].numpy(), map_ref[:, col].numpy()
            # col 0 (xyz): on this synthetic surface all xyz distances are ~0.1 of the unit feature scale, i.e.
            # inside the bf16 feature noise, so only an absolute bound is meaningful there
            assert np.abs(a - b).max() <= 0.05 * np.abs(b).max() + (0.1 if col == 0 else 0.02), (col, np.abs(a - b).max(), np.abs(b).max())
            if col == 1:
                assert np.corrcoef(a, b)[0, 1] > 0.99


@pytest.mark.parametrize("search_dtype,agree", [(torch.bfloat16, 0.9999), (torch.float16, 0.9999)], ids=["bf16", "fp16"])
def test_scoring_exact_features_isolated(fitted, search_dtype, agree, monkeypatch):
    """Same (oracle) features fed to the GPU scorer: isolates a11-a13 from the network tolerance -- on both operand types of the
    distance GEMM (bf16: the default since round 5; fp16: CMDIAD_SEARCH_DTYPE=fp16).  Rounds 1-5 took the 16-bit search's winner
    as it was: 4 of 784 near-ties went the other way with bf16 operands, 1-3 with fp16 (argmin agreement 99.5 %).  Since round 6
    the search also returns every query's runner-up and the fp32 re-score decides between the two (include/cmdiad_hip.h,
    cmdiad_l2_rescore2): min_idx is the float64 argmin on >= 99.99 % of the queries -- here: on all of them -- for both types.
    Uses the rgb modality: its patches are well separated (distances O(10)), whereas the xyz features of
    this smooth synthetic surface are near-duplicates whose distances sit below the 16-bit operand noise
    of ANY half-precision search (covered separately, with an absolute bound, in the predict test)."""
    from oracle import scoring
    cpu, cpu_feats, m, train = fitted
    monkeypatch.setattr(ops, "SEARCH_DTYPE", search_dtype)
    bank = eng.Bank(cpu.rgb_lib.to(DEV))
    assert bank.bf16.dtype == search_dtype
    # (1) a query that IS a bank row: the true distance is exactly 0 and the GPU path returns 0; the
    # reference's fp32 torch.cdist (matmul expansion) reports up to sqrt(|x|^2 * eps) ~ 0.03 there
    rp, _ = cpu_feats[1]
    q = (rp - cpu.rgb_mean) / cpu.rgb_std
    r = eng.score_patches(q.to(DEV).unsqueeze(0).contiguous(), bank, (28, 28))
    ref32 = scoring.single_s_s_map(q, torch.cdist(q, cpu.rgb_lib), cpu.rgb_lib, (28, 28), blur=False)
    assert float(r["min_val"][0].max()) < 1e-3
    np.testing.assert_allclose(r["min_val"][0].cpu().numpy(), ref32["min_val"].numpy(), atol=5e-2)
    assert (r["min_idx"][0].cpu() == ref32["min_idx"]).float().mean() > 0.995   # (ref32 = fp32 cdist at distance ~0: its own noise)
    # (2) an unseen query: every stage of compute_single_s_s_map against the same composition fed with a
    # float64 distance matrix (exact), and against the reference's fp32 cdist at its own error floor
    rgb, pc = synth_sample(12, True)
    rq, _ = cpu.ex(rgb, pc)
    q = (rq - cpu.rgb_mean) / cpu.rgb_std
    exact = torch.cdist(q.double(), cpu.rgb_lib.double()).float()
    ref = scoring.single_s_s_map(q, exact, cpu.rgb_lib, (28, 28), blur=False)
    ref32 = scoring.single_s_s_map(q, torch.cdist(q, cpu.rgb_lib), cpu.rgb_lib, (28, 28), blur=False)
    r = eng.score_patches(q.to(DEV).unsqueeze(0).contiguous(), bank, (28, 28))
    same = (r["min_idx"][0].cpu() == ref["min_idx"])
    assert same.float().mean() > agree, same.float().mean()
    np.testing.assert_allclose(r["min_val"][0].cpu().numpy()[same], ref["min_val"].numpy()[same], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r["min_val"][0].cpu().numpy(), ref["min_val"].numpy(), rtol=2e-3)  # flipped near-ties
    np.testing.assert_allclose(r["min_val"][0].cpu().numpy(), ref32["min_val"].numpy(), rtol=2e-3, atol=5e-2)
    assert int(r["s_idx"][0]) == int(ref["s_idx"])
    np.testing.assert_allclose(float(r["s_star"][0]), float(ref["s_star"]), rtol=1e-5)
    _, nn_idx = ops.unpack_keys(r["top3"][0])
    np.testing.assert_array_equal(nn_idx.cpu().numpy(), ref["nn_idx"].numpy())
    np.testing.assert_allclose(r["knn_d"][0].cpu().numpy(), ref["m_star_knn"].numpy(), rtol=1e-5)
    np.testing.assert_allclose(float(r["s"][0]), float(ref["s"]), rtol=1e-4)
    np.testing.assert_allclose(r["s_map_pre"][0].cpu().numpy(), ref["s_map_pre"][0].numpy(), rtol=2e-3, atol=1e-5)


def test_batch_invariance(weights):
    """B = 3 (ragged clouds, padded) gives the per-sample B = 1 results: indices and patch features bit for bit."""
    from cmdiad_amd import runtime
    sd_vit, sd_pm = weights
    e = eng.Engine(runtime.PackedViT(sd_vit, device=DEV), runtime.PackedPointMAE(sd_pm, device=DEV))
    samples = [synth_sample(20 + i) for i in range(3)]
    rgb = torch.cat([s[0] for s in samples]).to(DEV)
    pcs = torch.cat([s[1] for s in samples]).to(DEV)
    ex = e.extract(rgb, pcs)
    xp = e.xyz_patch(ex)
    rp = e.rgb_patch(ex)
    for i in range(3):
        ex1 = e.extract(rgb[i:i + 1], pcs[i:i + 1])
        n = int(ex1.n_valid[0])
        assert int(ex.n_valid[i]) == n
        assert torch.equal(ex.center_idx[i], ex1.center_idx[0]) and torch.equal(ex.ori_idx[i], ex1.ori_idx[0])
        assert torch.equal(ex.idx3[i, :n], ex1.idx3[0, :n])
        assert torch.equal(e.xyz_patch(ex1)[0], xp[i]) and torch.equal(e.rgb_patch(ex1)[0], rp[i])      # bit for bit


def test_greedy_coreset_matches_fp16_restatement():
    """cmdiad_coreset_greedy vs a torch-CPU restatement of features.py:372-425 with the fp16 semantics
    stated in coreset.hip (difference in fp16, fp32 accumulate, fp16 result, first max)."""
    from cmdiad_amd import coreset
    g = torch.Generator().manual_seed(7)
    z = torch.randn(3000, 62, generator=g)
    sel = coreset.greedy_coreset(z.to(DEV), 200).cpu()
    zh = z.half()
    min_d = torch.linalg.norm(z - z[0:1], dim=1).half()
    ref = [0]
    last = zh[0:1]
    for _ in range(199):
        d = torch.linalg.norm((zh - last).float(), dim=1).half()
        min_d = torch.minimum(d, min_d)
        mx = min_d.max()
        i = int(torch.nonzero(min_d == mx)[0])
        ref.append(i)
        last = zh[i:i + 1]
    np.testing.assert_array_equal(sel.numpy(), np.array(ref))


def test_coreset_early_termination_keeps_every_pick(monkeypatch):
    """The partial-distance early exit of the round kernel (a wave stops reading its rows' dimensions once every row's partial
    distance has reached its current minimum) is exact: same picks as the full scan on clustered rows with exact duplicates
    (the structure of a patch library: background patches repeat, neighbouring patches are close)."""
    from cmdiad_amd import coreset
    g = torch.Generator().manual_seed(17)
    centres = torch.randn(40, 334, generator=g) * 3
    z = centres[torch.randint(0, 40, (30000,), generator=g)] + 0.3 * torch.randn(30000, 334, generator=g)
    z[5000:9000] = z[4999]                 # a run of identical rows
    z[20000:20300] = 0.0
    picks = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CMDIAD_CORESET_EARLY", mode)
        picks[mode] = coreset.greedy_coreset(z.to(DEV), 1500).cpu()
    assert torch.equal(picks["1"], picks["0"]) and len(set(picks["1"].tolist())) == 1500


def test_coreset_vs_reference_golden(golden):
    """get_coreset_idx_randomp (host sparse projection + cmdiad_coreset_greedy) against the selection the REFERENCE's
    own function made on the same rows (tests/golden/g9_coreset.npz, features.py:360-425)."""
    from cmdiad_amd.feature_extractors.features import Features
    g = golden("g9_coreset.npz")
    z = torch.randn(int(g["rows"]), int(g["dim"]), generator=torch.Generator().manual_seed(int(g["z_seed"])))
    fake = types.SimpleNamespace(args=types.SimpleNamespace(dist_method_coreset="l2"), random_state=int(g["random_state"]), device=DEV)
    sel = Features.get_coreset_idx_randomp(fake, z, n=int(g["n"]), eps=float(g["eps"]), coreset_dtype="FP16")
    np.testing.assert_array_equal(sel.numpy(), g["idx"])


def test_coreset_tf32_vs_reference_golden(golden):
    """coreset_dtype='TF32' (features.py:390-391, main.py:151 -- in the reference an fp32 scan): cmdiad_coreset_greedy_f32 against
    the selection the reference's own function made with that setting (tests/golden/g9b_coreset_tf32.npz), and against a float64
    restatement of the greedy rule on a larger set (every pick attains the largest running minimum to fp32 accuracy)."""
    from cmdiad_amd import coreset
    from cmdiad_amd.feature_extractors.features import Features
    g = golden("g9b_coreset_tf32.npz")
    z = torch.randn(int(g["rows"]), int(g["dim"]), generator=torch.Generator().manual_seed(int(g["z_seed"])))
    fake = types.SimpleNamespace(args=types.SimpleNamespace(dist_method_coreset="l2"), random_state=int(g["random_state"]), device=DEV)
    sel = Features.get_coreset_idx_randomp(fake, z, n=int(g["n"]), eps=float(g["eps"]), coreset_dtype="TF32")
    np.testing.assert_array_equal(sel.numpy(), g["idx"])
    zz = torch.randn(30001, 333, generator=torch.Generator().manual_seed(5))          # odd d, n % 4 != 0
    picks = coreset.greedy_coreset(zz.to(DEV), 400, "TF32").cpu()
    assert picks[0] == 0 and len(set(picks.tolist())) == 400
    z64 = zz.double()
    min_d = torch.linalg.norm(z64 - z64[0:1], dim=1)
    for i in picks[1:].tolist():                   # the greedy rule in float64: each pick is an arg-max of the running minimum
        assert float(min_d[i]) >= float(min_d.max()) * (1 - 1e-5)
        min_d = torch.minimum(min_d, torch.linalg.norm(z64 - z64[i:i + 1], dim=1))
    with pytest.raises(NotImplementedError):
        coreset.greedy_coreset(zz.to(DEV), 4, "BF16")


@pytest.mark.parametrize("n,d,eps,seed", [(5003, 768, 0.9, 3), (20000, 1152, 0.9, 0), (777, 256, 0.95, 11), (9, 2048, 0.9, 5)])
def test_sparse_random_projection_on_device_is_bit_identical_to_sklearn(n, d, eps, seed):
    """features.py:360-363: SparseRandomProjection(eps, random_state).fit_transform on the host against
    coreset.sparse_random_projection (scikit-learn fits, cmdiad_sparse_project_f32 transforms): same shape, same bits."""
    from sklearn import random_projection
    from cmdiad_amd import coreset
    z = torch.randn(n, d, generator=torch.Generator().manual_seed(n + d))
    want = random_projection.SparseRandomProjection(eps=eps, random_state=seed).fit_transform(z.numpy())
    got = coreset.sparse_random_projection(z.to(DEV), eps, seed)
    assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_sparse_random_projection_raises_like_sklearn_when_eps_is_too_small():
    from cmdiad_amd import coreset
    with pytest.raises(ValueError):   # the Johnson-Lindenstrauss dimension for eps = 0.1 exceeds 64 features
        coreset.sparse_random_projection(torch.randn(1000, 64).to(DEV), 0.1, 0)


def test_late_fusion_fit_on_device_equals_sklearn(weights, monkeypatch):
    """features.py:352-358 with CMDIAD_OCSVM_DEVICE=1: the drop-in's two one-class SVMs are fitted by cmdiad_ocsvm_fit and equal
    scikit-learn fitted on the same s_lib / s_map_lib rows bit for bit (score maps of 2 train samples: 100 352 rows)."""
    import warnings
    from sklearn import linear_model
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures
    from cmdiad_amd.ocsvm import DeviceSGDOneClassSVM
    sd_vit, sd_pm = weights
    monkeypatch.setenv("CMDIAD_OCSVM_DEVICE", "1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DoubleRGBPointFeatures(make_args())
    assert isinstance(m.detect_fuser, DeviceSGDOneClassSVM) and isinstance(m.seg_fuser, DeviceSGDOneClassSVM)
    m.deep_feature_extractor.rgb_backbone.load_state_dict(sd_vit)
    m.deep_feature_extractor.xyz_backbone.load_state_dict(sd_pm)
    train = [synth_sample(i) for i in range(2)]
    for rgb, pc in train:
        m.add_sample_to_mem_bank((rgb, pc, pc), class_name="synth")
    m.run_coreset()
    for rgb, pc in train:
        m.add_sample_to_late_fusion_mem_bank((rgb, pc, pc))
    m.run_late_fusion()
    assert m.s_map_lib.dtype == torch.float32 and m.s_map_lib.shape == (2 * 50176, 2)
    det = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(m.s_lib)
    seg = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(m.s_map_lib)
    for got, ref in ((m.detect_fuser, det), (m.seg_fuser, seg)):
        np.testing.assert_array_equal(got.coef_, ref.coef_)
        np.testing.assert_array_equal(got.offset_, ref.offset_)
        assert got.n_iter_ == ref.n_iter_


def test_full_protocol_double_rgb_point_vs_oracle(fitted):
    """cmdiad_runner.py:44-92 end to end for DINO+Point_MAE: late-fusion bank, OCSVM fit, predict, metrics --
    the drop-in against the CPU oracle driven through the same protocol (same scikit-learn on both sides)."""
    from sklearn import linear_model
    cpu, cpu_feats, m, train = fitted
    # ---- GPU drop-in: pass 2 + late fusion
    for rgb, pc in train:
        m.add_sample_to_late_fusion_mem_bank((rgb, pc, pc))
    m.run_late_fusion()
    # ---- CPU oracle: the same (multiple_features.py:897-927, features.py:352-358)
    s_lib, s_map_lib = [], []
    for rp, xp in cpu_feats:
        s, s_map, _, _ = cpu.score(rp, xp)
        s_lib.append(s)
        s_map_lib.append(s_map)
    det = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(torch.cat(s_lib, 0))
    seg = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(torch.cat(s_map_lib, 0))
    # SGD on 4 rows is ill-conditioned (its solution flips with 1e-3 input changes), so both sides score with the
    # SAME fitted linear models: this pins the protocol glue (stacking, lambdas, score_samples plumbing, metrics)
    assert m.detect_fuser.coef_.shape == det.coef_.shape and m.seg_fuser.coef_.shape == seg.coef_.shape
    m.detect_fuser, m.seg_fuser = det, seg
    tests = [(30, False), (31, True), (32, False), (33, True)]
    ref_img, ref_pix = [], []
    for i, anomalous in tests:
        rgb, pc = synth_sample(i, anomalous)
        mask = torch.zeros(1, 1, 224, 224)
        if anomalous:
            mask[..., 100:120, 100:120] = 1
        m.predict((rgb, pc, pc), mask, np.array([int(anomalous)]), [f"synth/{i}.png"])
        s, s_map, _, _ = cpu.predict(rgb, pc)
        ref_img.append(det.score_samples(s))
        ref_pix.append(seg.score_samples(s_map))
    m.calculate_metrics()
    got_img = np.concatenate(m.image_preds).ravel()
    ref_img = np.concatenate(ref_img).ravel()
    assert np.isfinite([m.image_rocauc, m.pixel_rocauc, m.au_pro, m.au_pro_001]).all()
    # the planted RGB anomaly (+2 sigma patch) dominates the image score on both sides: same ranking
    assert list(np.argsort(got_img)) == list(np.argsort(ref_img)), (got_img, ref_img)
    spread = float(ref_img.max() - ref_img.min())
    assert np.abs(got_img - ref_img).max() <= 0.35 * spread + 1e-6, (got_img, ref_img)
    from sklearn.metrics import roc_auc_score
    labels = np.array([int(a) for _, a in tests])
    assert np.isfinite(roc_auc_score(labels, got_img))
    ref_pix_auc = roc_auc_score(np.array(m.pixel_labels).astype(int), np.concatenate(ref_pix))
    assert abs(m.pixel_rocauc - ref_pix_auc) < 0.03, (m.pixel_rocauc, ref_pix_auc)


def test_other_method_classes_run_the_protocol(weights):
    """RGBFeatures, PointFeatures (with the greedy coreset, f_coreset = 0.5) and the FtoF hallucination class
    run fit + late fusion + predict + metrics on the GPU path and produce finite, well-formed results."""
    import warnings
    from cmdiad_amd.feature_extractors import multiple_features as mf
    sd_vit, sd_pm = weights
    train = [synth_sample(40 + i) for i in range(3)]
    tests = [(50, False), (51, True)]
    for cls, kw in ((mf.RGBFeatures, {}), (mf.PointFeatures, dict(f_coreset=0.5, random_state=0)),
                    (mf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality='xyz'))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = cls(make_args(**kw))
        m.deep_feature_extractor.rgb_backbone.load_state_dict(sd_vit)
        m.deep_feature_extractor.xyz_backbone.load_state_dict(sd_pm)
        if kw.get("use_hn"):
            m.fusion.load_state_dict(nets.synth_state_dict("halluc", 51))
        for rgb, pc in train:
            m.add_sample_to_mem_bank((rgb, pc, pc), class_name="synth")
        m.run_coreset()
        if cls is mf.PointFeatures:
            assert m.patch_xyz_lib.shape == (int(0.5 * 3 * 3136), 768)
            # (background patches are exact duplicates -- all-zero features -- so once every remaining row is at
            # distance 0 the greedy argmax returns row 0 again, exactly as the reference's loop does; the
            # selection itself is pinned by test_greedy_coreset_matches_fp16_restatement)
            idx = m.coreset_idx
            assert int(idx[0]) == 0 and int(idx.min()) >= 0 and int(idx.max()) < 3 * 3136
            assert len(set(idx.tolist())) > 1000
        for rgb, pc in train:
            m.add_sample_to_late_fusion_mem_bank((rgb, pc, pc))
        m.run_late_fusion()
        for i, anomalous in tests:
            rgb, pc = synth_sample(i, anomalous)
            mask = torch.zeros(1, 1, 224, 224)
            if anomalous:
                mask[..., 100:120, 100:120] = 1
            m.predict((rgb, pc, pc), mask, np.array([int(anomalous)]), [f"synth/{i}.png"])
        m.calculate_metrics()
        assert np.isfinite([m.image_rocauc, m.pixel_rocauc, m.au_pro]).all(), cls.__name__
        assert m.predictions[0].shape == (224, 224)


def test_depth_features_is_the_rgb_method_on_the_third_sample_slot(weights):
    """multiple_features.py:124-200: DepthFeatures runs the RGB backbone on sample[2] (the three-channel depth image).  Given the
    photograph in slot 2 (and garbage in slot 0) it returns what RGBFeatures returns for the photograph in slot 0, bit for bit;
    Features.interpolate_points (features.py:216-219) returns the point branch of one forward pass."""
    import warnings
    from cmdiad_amd.feature_extractors import multiple_features as mf
    sd_vit, sd_pm = weights
    train = [synth_sample(40 + i) for i in range(3)]
    outs = []
    for cls, slot in ((mf.RGBFeatures, 0), (mf.DepthFeatures, 2)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = cls(make_args())
        m.deep_feature_extractor.rgb_backbone.load_state_dict(sd_vit)
        m.deep_feature_extractor.xyz_backbone.load_state_dict(sd_pm)
        place = lambda rgb, pc: (rgb, pc, torch.full_like(rgb, 7.0)) if slot == 0 else (torch.full_like(rgb, 7.0), pc, rgb)   # noqa: E731
        for rgb, pc in train:
            m.add_sample_to_mem_bank(place(rgb, pc), class_name="synth")
        m.run_coreset()
        for rgb, pc in train:
            m.add_sample_to_late_fusion_mem_bank(place(rgb, pc))
        m.run_late_fusion()
        for i, anomalous in ((50, False), (51, True)):
            rgb, pc = synth_sample(i, anomalous)
            m.predict(place(rgb, pc), torch.zeros(1, 1, 224, 224), np.array([int(anomalous)]), [f"synth/{i}.png"])
        outs.append((np.concatenate([np.ravel(x) for x in m.image_preds]), np.stack(m.predictions)))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    rgb, pc = train[0]
    pts, _ = mf.organized_pc_to_unorganized_pc_no_zeros((rgb, pc, pc))
    xyz_maps, center, xyz_back = m.interpolate_points(rgb, pts.contiguous())
    full = m(rgb, pts.contiguous())
    assert torch.equal(xyz_maps[0], full[1][0]) and torch.equal(center.cpu(), full[2].cpu()) and xyz_back is not None


def test_public_features_contract(weights):
    """the PUBLIC surface the reference's callers use -- Features.__call__(rgb, xyz_unorganized) for all
    three out_types (features.py:123-158), get_xyz_patch with CALLER-supplied nonzero_indices (:169-184),
    LazyInterpolated.materialize() against interpolating_points (pointnet2_utils.py:45-75), Model.forward
    (models.py:55-67), calculate_dist -> DistHandle.materialize() / .min(1) (features.py:186-190, 227)."""
    import warnings
    from cmdiad_amd.feature_extractors.features import DistHandle, LazyInterpolated
    from cmdiad_amd.feature_extractors.multiple_features import DoubleRGBPointFeatures, organized_pc_to_unorganized_pc_no_zeros
    from oracle import scoring
    sd_vit, sd_pm = weights
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DoubleRGBPointFeatures(make_args())
    m.deep_feature_extractor.rgb_backbone.load_state_dict(sd_vit)
    m.deep_feature_extractor.xyz_backbone.load_state_dict(sd_pm)
    rgb, pc = synth_sample(60)
    sample = (rgb, pc, pc)
    xyz, nz = organized_pc_to_unorganized_pc_no_zeros(sample)
    ref_pc, ref_nz = scoring.unorganize_no_zeros(pc)
    assert torch.equal(xyz, ref_pc) and np.array_equal(nz, ref_nz)
    cpu = CpuExtractor(sd_vit, sd_pm)
    ref_rgb_patch, ref_xyz_patch = cpu(rgb, pc)

    # ---- out_type = "rgb+xyz": the reference's 6-tuple, feature maps on the CPU
    rgb_maps, xyz_maps, center, ori_idx, center_idx, interp = m(rgb, xyz.contiguous())
    assert isinstance(rgb_maps, list) and rgb_maps[0].device.type == "cpu" and tuple(rgb_maps[0].shape) == (1, 768, 28, 28)
    assert isinstance(xyz_maps, list) and xyz_maps[0].device.type == "cpu" and tuple(xyz_maps[0].shape) == (1, 768, 1024)
    assert tuple(center.shape) == (1, 1024, 3) and tuple(ori_idx.shape) == (1, 1024, 128) and tuple(center_idx.shape) == (1, 1024)
    assert isinstance(interp, LazyInterpolated) and tuple(interp.shape) == (1, 768, xyz.shape[2])
    # FPS / kNN indices are bit-exact against the oracle
    from oracle import kernels as ok
    pts = np.ascontiguousarray(xyz[0].T.numpy())[None]
    ridx, rcen = ok.fps(pts, 1024)
    assert np.array_equal(center_idx.cpu().numpy(), ridx) and np.array_equal(center.cpu().numpy(), rcen)
    rk, _ = ok.knn_group(pts, rcen, 128)
    assert np.array_equal(ori_idx.cpu().numpy(), rk)
    # feature maps vs the fp32 oracle (bf16 network tolerance)
    mr, xr = _rel(rgb_maps[0].reshape(768, -1).T, ref_rgb_patch)
    assert mr < 0.02 and xr < 0.2, (mr, xr)
    # LazyInterpolated.materialize() == interpolating_points on the SAME (GPU) centre features
    full = interp.materialize().cpu()
    want = scoring.interpolating_points(xyz, center.cpu().permute(0, 2, 1), xyz_maps[0])
    err = (full - want).abs()
    # (-2ab + a^2 + b^2 cancels for the ~4 % of points that ARE centres: their weights depend on the summation order)
    assert err.mean() < 2e-3 * want.abs().mean() + 1e-4 and (err > 0.05 * want.abs().mean()).float().mean() < 0.06
    assert torch.equal(interp.to("cpu"), full)
    # get_xyz_patch with the caller's nonzero_indices (no pix2pt from a device-side unorganise on this path)
    assert interp.ex.pix2pt is None
    xp = m.get_xyz_patch(xyz_maps, interp, nz)
    assert tuple(xp.shape) == (3136, 768)
    mx, xx = _rel(xp.cpu(), ref_xyz_patch)
    assert mx < 0.03 and xx < 0.3, (mx, xx)
    xp28 = m.get_xyz_patch(xyz_maps, interp, nz, get_2828=True)
    ref28 = scoring.get_xyz_patch(want, nz, out=28)
    assert tuple(xp28.shape) == (784, 768)
    assert _rel(xp28.cpu(), ref28)[0] < 0.03
    # the method classes' private device path gives the same patches
    ex = m._extract_device(rgb, pc)
    assert torch.equal(m._engine.xyz_patch(ex)[0].cpu(), xp.cpu())
    rp, rp2 = m.get_rgb_patch(rgb_maps)
    assert tuple(rp.shape) == (784, 768) and tuple(rp2.shape) == (3136, 768)
    assert torch.equal(rp.cpu(), m._engine.rgb_patch(ex)[0].cpu())
    # get_rgb_patch on a plain CPU tensor list (no device handle attached), as a caller that re-built the list would pass
    rp_plain, rp2_plain = m.get_rgb_patch([rgb_maps[0].clone()])
    torch.testing.assert_close(rp_plain, rp, rtol=0, atol=0)
    ref_p, ref_p2 = scoring.get_rgb_patch(rgb_maps[0])
    torch.testing.assert_close(rp2_plain.cpu(), ref_p2, rtol=0, atol=0)

    # ---- out_type = "rgb" and "xyz"
    only_rgb = m(rgb=rgb, out_type="rgb")
    assert isinstance(only_rgb, list) and torch.equal(only_rgb[0], rgb_maps[0])
    xyz_maps2, center2, ori2, cidx2, interp2 = m(xyz=xyz.contiguous(), out_type="xyz")
    assert torch.equal(xyz_maps2[0], xyz_maps[0]) and torch.equal(center2, center) and torch.equal(ori2, ori_idx)
    assert torch.equal(cidx2, center_idx) and isinstance(interp2, LazyInterpolated)

    # ---- Model.forward (models.py:55-67): device tensors in the reference's layouts
    dfe = m.deep_feature_extractor
    f_rgb, f_xyz, c3, o3, ci3 = dfe(rgb.cuda(), xyz.cuda().contiguous())
    assert tuple(f_rgb.shape) == (1, 768, 28, 28) and tuple(f_xyz.shape) == (1, 768, 1024)
    assert torch.equal(f_rgb.cpu(), rgb_maps[0]) and torch.equal(f_xyz.cpu(), xyz_maps[0]) and torch.equal(ci3, center_idx)
    assert torch.equal(dfe(rgb=rgb.cuda(), out_type="rgb").cpu(), rgb_maps[0])
    assert torch.equal(dfe(xyz=xyz.cuda().contiguous(), out_type="xyz")[0].cpu(), xyz_maps[0])

    # ---- calculate_dist: handle, exact materialisation, min without the matrix, and the library guard
    lib = torch.randn(1500, 768, generator=torch.Generator().manual_seed(3)).cuda()
    q = (rp - rp.mean()) / rp.std()
    h = m.calculate_dist(q, lib)
    assert isinstance(h, DistHandle) and tuple(h.shape) == (784, 1500)
    dm = h.materialize()
    ref_d = torch.cdist(q.double().cpu(), lib.double().cpu())
    np.testing.assert_allclose(dm.cpu().numpy(), ref_d.numpy(), rtol=2e-6, atol=1e-5)
    mv, mi = h.min(1)
    rmv, rmi = ref_d.min(1)
    assert (mi.cpu() == rmi).float().mean() > 0.99      # (bf16 search operands: 4 of 784 near-ties flip; their distances: next line)
    np.testing.assert_allclose(mv.cpu().numpy(), rmv.numpy(), rtol=2e-3)
    m.patch_rgb_lib = lib
    s, s_map = m.compute_single_s_s_map(q, h, (28, 28), modal="rgb")
    ref = scoring.single_s_s_map(q.cpu(), ref_d.float(), lib.cpu(), (28, 28))
    np.testing.assert_allclose(float(s), float(ref["s"]), rtol=1e-3)
    assert tuple(s_map.shape) == (1, 224, 224) and np.abs(s_map.numpy() - ref["s_map"].numpy()).max() <= float(ref["s_map"].max()) / 255 * 1.01
    m.patch_xyz_lib = torch.randn(10, 768).cuda()
    with pytest.raises(ValueError, match="different library"):
        m.compute_single_s_s_map(q, h, (28, 28), modal="xyz")
