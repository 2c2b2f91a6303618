"""GPU: the batched predict of the package (cmdiad_amd.predictor.BatchPredictor = engine.predict_batch, what bench.py times)
and the single-library / MTFI drop-in classes, end to end against the CPU oracle pipelines (oracle/pipeline.py, pinned to
the reference's own classes by goldens G6 / G11) and against the reference's outputs themselves (G11).

The synthetic class uses oracle.nets.sharpen_pointmae weights: with them the xyz patch-to-patch distances are ~4 (ten
times the bf16 feature error), so BOTH modalities are checked relatively -- round 1's smooth-surface class left the xyz
column inside the bf16 noise."""
import types
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import engine as eng  # noqa: E402
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd import runtime  # noqa: E402
from cmdiad_amd.predictor import BatchPredictor  # noqa: E402
from cmdiad_amd.synth import synth_cloud, synth_rgb  # noqa: E402
from oracle import nets, pipeline  # noqa: E402
import functools  # noqa: E402

from conftest import pmap, prefetched  # noqa: E402

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


def synth_sample(i, anomalous=False, frac=None):
    pc = synth_cloud(500 + i, frac or (0.40 + 0.03 * (i % 4)), texture=0.004)
    rgb = synth_rgb(500 + i)
    mask = torch.zeros(1, 224, 224)
    if anomalous:
        y0, x0 = 70 + 9 * (i % 7), 80 + 7 * (i % 5)
        # SURVEY 8d plants a 5 mm dent and a +2 sigma colour shift; on THIS class (every image is iid noise, 4 mm relief) that
        # leaves the image-level scores of normal and anomalous samples interleaved (I-AUROC 0.70), where a 2 % score
        # difference swaps ranks.  A 15 mm dent and +4 sigma give a margin, so I-AUROC parity measures the scorer, not luck.
        pc[0, 2, y0:y0 + 20, x0:x0 + 20] -= 0.015 * (pc[0, 2, y0:y0 + 20, x0:x0 + 20] != 0)
        rgb[0, :, y0:y0 + 20, x0:x0 + 20] += 4.0
        mask[0, y0:y0 + 20, x0:x0 + 20] = 1
    return rgb, pc, mask


@functools.lru_cache(maxsize=None)
def _weights():
    return (nets.synth_state_dict("vit", 31), nets.sharpen_pointmae(nets.synth_state_dict("pointmae", 21)),
            nets.synth_state_dict("halluc", 51))


@functools.lru_cache(maxsize=None)
def _cpu_ex():
    w = _weights()
    return pipeline.CpuExtractor(w[0], w[1])


@pytest.fixture(scope="module")
def weights():
    return _weights()


@pytest.fixture(scope="module")
def cpu_ex():
    return _cpu_ex()


@pytest.fixture(scope="module")
def gpu_engine(weights):
    return eng.Engine(runtime.PackedViT(weights[0], device=DEV), runtime.PackedPointMAE(weights[1], device=DEV))


def _fit_svms(rows_s, rows_map):
    from sklearn import linear_model
    det = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(torch.cat(rows_s, 0).numpy())
    seg = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(torch.cat(rows_map, 0)[::7].numpy())
    return det, seg


def _auroc_pair(labels, got, ref):
    from sklearn.metrics import roc_auc_score
    return roc_auc_score(labels, got), roc_auc_score(labels, ref)


def test_feature_error_is_far_below_the_patch_distances(weights, cpu_ex, gpu_engine):
    """The premise of this file: on the sharpened synthetic class the GPU (bf16) patch features differ from the fp32
    oracle's by much less than the nearest-neighbour distances the scorer measures (a test sample's patches against
    ANOTHER sample's patches).  Measured on MI355X: xyz |err| 0.51 vs distance 3.9 (round 1's class: 0.4 vs 0.02);
    rgb 0.13 vs 19."""
    rgb, pc, _ = synth_sample(0)
    rp, xp = cpu_ex(rgb, pc)
    rp_b, xp_b = cpu_ex(*synth_sample(1)[:2])
    ex = gpu_engine.extract(rgb.to(DEV), pc.to(DEV))
    gx = gpu_engine.xyz_patch(ex)[0].cpu()
    gr = gpu_engine.rgb_patch(ex)[0].cpu()
    fg = xp.abs().sum(1) > 0
    err_x = (gx - xp)[fg].norm(dim=1)
    err_r = (gr - rp).norm(dim=1)
    nn_x = torch.cdist(xp[fg], xp_b).min(1).values
    nn_r = torch.cdist(rp, rp_b).min(1).values
    print(f"xyz: |err| mean {err_x.mean():.3f} max {err_x.max():.3f}; nn dist mean {nn_x.mean():.3f}; "
          f"rgb: |err| mean {err_r.mean():.3f}; nn dist mean {nn_r.mean():.3f}")
    assert err_x.mean() < 0.2 * nn_x.mean() and err_r.mean() < 0.05 * nn_r.mean()


def oracle_b32():
    """The CPU side of test_predict_batch_b32_vs_oracle (no GPU call): oracle fit on 4 train samples, late-fusion models from 3
    further normal samples, and the oracle's predict of the 32 test samples."""
    cpu = pipeline.CpuDoubleRGBPoint(_cpu_ex())
    train = [synth_sample(100 + i)[:2] for i in range(4)]
    cpu.fit(train)
    # late-fusion models from the oracle's scores of 3 further normal samples (shared by both sides: only coef_/offset_ matter)
    rows = [cpu.predict(*synth_sample(200 + i)[:2])[:2] for i in range(3)]
    det, seg = _fit_svms([r[0] for r in rows], [r[1] for r in rows])
    B = 32
    samples = [synth_sample(i, anomalous=(i % 3 == 0 and i < 30)) for i in range(B)]
    labels = np.array([int(s[2].any()) for s in samples])
    assert labels.sum() == 10
    ref = pmap(lambda smp: cpu.predict(smp[0], smp[1])[:2], samples, 8)       # the oracle, eight samples at a time on the host cores
    ref_img = np.array([float(det.score_samples(s.numpy())[0]) for s, _ in ref])
    ref_pix = np.stack([seg.score_samples(s_map.numpy()).reshape(224, 224) for _, s_map in ref])
    return cpu, det, seg, samples, labels, ref_img, ref_pix


@pytest.mark.oracle_prefetch("oracle_b32")
def test_predict_batch_b32_vs_oracle(gpu_engine, monkeypatch):
    """B = 32 through BatchPredictor (HIP graphs, both buffer sets, and the eager path) against
    oracle.pipeline.CpuDoubleRGBPoint.predict sample by sample: image score, blurred pixel map, and I-/P-AUROC over the 32
    samples (10 anomalous)."""
    cpu, det, seg, samples, labels, ref_img, ref_pix = prefetched(oracle_b32)
    B = len(samples)

    bank_xyz, bank_rgb = eng.Bank(cpu.xyz_lib.to(DEV)), eng.Bank(cpu.rgb_lib.to(DEV))
    stats = dict(xyz_mean=float(cpu.xyz_mean), xyz_std=float(cpu.xyz_std), rgb_mean=float(cpu.rgb_mean), rgb_std=float(cpu.rgb_std))
    rgb = torch.cat([s[0] for s in samples]).to(DEV)
    pcs = torch.cat([s[1] for s in samples]).to(DEV)
    outs = {}
    for mode in ("graph", "eager"):
        p = BatchPredictor(gpu_engine, bank_xyz, bank_rgb, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=B,
                           use_graph=(mode == "graph"))
        first = p.predict_batch(rgb, pcs)
        second = p.predict_batch(rgb.cpu().pin_memory(), pcs.cpu().pin_memory())   # other buffer set, host-fed
        assert mode != "graph" or p.use_graph, "HIP graph capture failed"
        assert np.array_equal(first[0], second[0]) and np.array_equal(first[1], second[1])
        outs[mode] = first
    assert np.array_equal(outs["graph"][0], outs["eager"][0]) and np.array_equal(outs["graph"][1], outs["eager"][1])
    # the exact removal of the repeated background rows in front of the xyz search (csrc/dedup.hip) changes no output bit, and
    # it did search fewer rows (these clouds cover 40-49 % of the image)
    assert p.dedup and 0 < int(p.live_rows.item()) < 0.8 * B * 3136 * p.xyz_searches, (int(p.live_rows.item()), p.xyz_searches)
    monkeypatch.setenv("CMDIAD_DEDUP", "0")
    p_all = BatchPredictor(gpu_engine, bank_xyz, bank_rgb, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=B, use_graph=False)
    monkeypatch.delenv("CMDIAD_DEDUP")
    every = p_all.predict_batch(rgb, pcs)
    assert not p_all.dedup and int(p_all.live_rows.item()) == B * 3136 * p_all.xyz_searches
    assert np.array_equal(every[0], outs["graph"][0]) and np.array_equal(every[1], outs["graph"][1])
    # the searches on the MAIN stream (CMDIAD_SEARCH_POST=0; the default runs them on the second stream, beside the next step's
    # extraction): same outputs
    monkeypatch.setenv("CMDIAD_SEARCH_POST", "0")
    p_post = BatchPredictor(gpu_engine, bank_xyz, bank_rgb, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=B, use_graph=True)
    tickets = [p_post.submit(rgb, pcs) for _ in range(2)]      # two steps in flight: both buffer sets, searches overlapping stage 1
    monkeypatch.delenv("CMDIAD_SEARCH_POST")
    for t in tickets:
        got = t.wait()
        assert np.array_equal(got[0], outs["graph"][0]) and np.array_equal(got[1], outs["graph"][1])
    img, pix = outs["graph"]
    assert img.shape == (B,) and pix.shape == (B, 224, 224) and img.dtype == np.float64

    spread = float(ref_img.max() - ref_img.min())
    d_img = np.abs(img - ref_img)
    d_pix = np.abs(pix - ref_pix).reshape(B, -1)
    print(f"image score: max |d| {d_img.max():.4f} of spread {spread:.4f}; pixel map: max |d| {d_pix.max():.5f}, "
          f"mean {d_pix.mean():.6f}, map range {np.ptp(ref_pix):.4f}")
    # image score = detect_fuser over [s_xyz, 0.1 s_rgb], s = w * max_q min_n d(q, n): the maximum of 3136 distances picks
    # up the bf16 feature noise in quadrature (d ~ 4, |err| ~ 0.5 -> +0.8 %), measured <= 2.3 % of the score = 7 % of the
    # (narrow) spread between samples on this class
    assert d_img.max() <= 0.10 * spread and d_img.mean() <= 0.03 * spread, (d_img.max(), d_img.mean(), spread)
    assert (d_img / np.abs(ref_img)).max() <= 0.03, (d_img / np.abs(ref_img)).max()
    # measured on MI355X (round 3): mean 0.08 %, max 1.4 % of the map range
    assert d_pix.mean() <= 0.003 * np.ptp(ref_pix) and d_pix.max() <= 0.03 * np.ptp(ref_pix), (d_pix.mean(), d_pix.max(), np.ptp(ref_pix))
    for b in range(B):
        assert np.corrcoef(pix[b].ravel(), ref_pix[b].ravel())[0, 1] > 0.995, b
    i_got, i_ref = _auroc_pair(labels, img, ref_img)
    masks = np.stack([s[2].numpy().reshape(224, 224) for s in samples]).astype(int)
    p_got, p_ref = _auroc_pair(masks.ravel(), pix.ravel(), ref_pix.ravel())
    print(f"I-AUROC {i_got:.4f} (oracle {i_ref:.4f}); P-AUROC {p_got:.4f} (oracle {p_ref:.4f})")
    assert abs(i_got - i_ref) <= 1e-2 and abs(p_got - p_ref) <= 1e-2

    # per-sample result does not depend on the batch: the B = 1 drop-in composition gives the batch's numbers
    p1 = BatchPredictor(gpu_engine, bank_xyz, bank_rgb, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=1, use_graph=False)
    for b in (0, 13, 31):
        i1, m1 = p1.predict_batch(rgb[b:b + 1], pcs[b:b + 1])
        np.testing.assert_allclose(i1[0], img[b], rtol=1e-4, atol=1e-6)
        assert np.abs(m1[0] - pix[b]).max() <= 2.5 * np.ptp(pix[b]) / 255.0 + 1e-9   # at most one 8-bit blur level per column


def oracle_heavy_tailed():
    """The CPU side of test_predict_batch_heavy_tailed_weights_vs_oracle (no GPU call)."""
    w_vit, w_pm = nets.outlier_vit(31), nets.sharpen_pointmae(nets.outlier_pointmae(21))
    cpu_ex = pipeline.CpuExtractor(w_vit, w_pm)
    cpu = pipeline.CpuDoubleRGBPoint(cpu_ex)
    cpu.fit(pmap(lambda i: synth_sample(100 + i)[:2], range(4), 1))
    rows = pmap(lambda i: cpu.predict(*synth_sample(200 + i)[:2])[:2], range(3), 3)
    det, seg = _fit_svms([r[0] for r in rows], [r[1] for r in rows])
    B = 8
    samples = [synth_sample(i, anomalous=(i % 3 == 0)) for i in range(B)]
    ref = pmap(lambda smp: cpu.predict(smp[0], smp[1])[:2], samples, 8)
    ref_img = np.array([float(det.score_samples(s.numpy())[0]) for s, _ in ref])
    ref_pix = np.stack([seg.score_samples(s_map.numpy()).reshape(224, 224) for _, s_map in ref])
    return w_vit, w_pm, cpu, det, seg, samples, ref_img, ref_pix


@pytest.mark.oracle_prefetch("oracle_heavy_tailed")
def test_predict_batch_heavy_tailed_weights_vs_oracle():
    """test_predict_batch_b32_vs_oracle's protocol (fit, late-fusion models, batched predict against the
    oracle sample by sample) on heavy-tailed weights -- oracle.nets.outlier_vit (three residual channels at ~100x from block 2 on,
    one high-norm token) and outlier_pointmae (one 50x BatchNorm channel) on top of the sharpened Point-MAE -- at B = 8: the same
    score / map bounds as on the O(1) weights."""
    w_vit, w_pm, cpu, det, seg, samples, ref_img, ref_pix = prefetched(oracle_heavy_tailed)
    B = len(samples)
    gpu_engine = eng.Engine(runtime.PackedViT(w_vit, device=DEV), runtime.PackedPointMAE(w_pm, device=DEV))
    bank_xyz, bank_rgb = eng.Bank(cpu.xyz_lib.to(DEV)), eng.Bank(cpu.rgb_lib.to(DEV))
    stats = dict(xyz_mean=float(cpu.xyz_mean), xyz_std=float(cpu.xyz_std), rgb_mean=float(cpu.rgb_mean), rgb_std=float(cpu.rgb_std))
    rgb = torch.cat([s[0] for s in samples]).to(DEV)
    pcs = torch.cat([s[1] for s in samples]).to(DEV)
    p = BatchPredictor(gpu_engine, bank_xyz, bank_rgb, stats, det, seg, lambdas=(1.0, 1.0, 0.1, 0.1), batch=B, use_graph=False)
    img, pix = p.predict_batch(rgb, pcs)
    spread = float(ref_img.max() - ref_img.min())
    d_img = np.abs(img - ref_img)
    d_pix = np.abs(pix - ref_pix).reshape(B, -1)
    print(f"heavy-tailed weights: image score max |d| {d_img.max():.4f} of spread {spread:.4f} ({(d_img / np.abs(ref_img)).max():.4f} relative); "
          f"pixel map max |d| {d_pix.max():.5f}, mean {d_pix.mean():.6f}, map range {np.ptp(ref_pix):.4f}")
    assert (d_img / np.abs(ref_img)).max() <= 0.03, (d_img / np.abs(ref_img)).max()
    assert d_pix.mean() <= 0.003 * np.ptp(ref_pix) and d_pix.max() <= 0.03 * np.ptp(ref_pix), (d_pix.mean(), d_pix.max(), np.ptp(ref_pix))
    for b in range(B):
        assert np.corrcoef(pix[b].ravel(), ref_pix[b].ravel())[0, 1] > 0.995, b


def oracle_mtfi():
    """The CPU oracle's side of the MTFI batch test (no GPU call), computed once for both operand types of the search."""
    weights, cpu_ex = _weights(), _cpu_ex()
    cpu = pipeline.CpuOneHallucination(cpu_ex, weights[2], "xyz", lambdas=(1.0, 1.0, 1.0, 1.0))
    cpu.fit([synth_sample(100 + i)[:2] for i in range(4)])
    rows = [cpu.predict(*synth_sample(200 + i)[:2])[:2] for i in range(2)]
    det, seg = _fit_svms([r[0] for r in rows], [r[1] for r in rows])
    B = 24
    samples = [synth_sample(i, anomalous=(i % 3 == 0)) for i in range(B)]
    labels = np.array([int(s[2].any()) for s in samples])
    assert labels.sum() == 8
    ref = pmap(lambda smp: cpu.predict(smp[0], smp[1])[:2], samples, 8)
    ref_img = np.array([float(det.score_samples(s.numpy())[0]) for s, _ in ref])
    ref_pix = np.stack([seg.score_samples(s_map.numpy()).reshape(224, 224) for _, s_map in ref])
    return cpu, det, seg, samples, labels, ref_img, ref_pix


@pytest.mark.oracle_prefetch("oracle_mtfi")
@pytest.mark.parametrize("search_dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_mtfi_batch_vs_oracle(search_dtype, weights, gpu_engine, monkeypatch):
    """`--workload mtfi` (configs[4] per-GPU work): BatchPredictor with the hallucination network against
    oracle.pipeline.CpuOneHallucination(main xyz).predict -- [xyz, hallucinated-rgb] columns, cross-wired statistics -- at
    B = 24 (8 anomalous): image scores, pixel maps, and I-/P-AUROC over the batch; on both operand types of the library search
    (bf16: the default; fp16: what BASELINE configs[4] names, CMDIAD_SEARCH_DTYPE=fp16)."""
    cpu, det, seg, samples, labels, ref_img, ref_pix = prefetched(oracle_mtfi)
    B = len(samples)
    monkeypatch.setattr(ops, "SEARCH_DTYPE", search_dtype)
    stats = dict(xyz_mean=float(cpu.mean), xyz_std=float(cpu.std), rgb_mean=float(cpu.mean), rgb_std=float(cpu.std))
    p = BatchPredictor(gpu_engine, eng.Bank(cpu.main_lib.to(DEV)), eng.Bank(cpu.fus_lib.to(DEV)), stats, det, seg,
                       lambdas=(1.0, 1.0, 1.0, 1.0), batch=B, workload="mtfi",
                       halluc=runtime.PackedHallucination(weights[2], device=DEV))
    assert p.bank_xyz.bf16.dtype == search_dtype
    pcs = torch.cat([s[1] for s in samples]).to(DEV)
    img, pix = p.predict_batch(None, pcs)
    img2, pix2 = p.predict_batch(None, pcs)
    assert p.use_graph and np.array_equal(img, img2) and np.array_equal(pix, pix2)
    # the distillation MLP and both searches ran on the de-duplicated rows only (one plan on the fp32 bit patterns of the raw xyz
    # patches, GEMMs sized by the device-side row count): fewer rows, the same bits as running every row
    assert p.dedup and 0 < int(p.live_rows.item()) < 0.8 * B * 3136 * p.xyz_searches
    monkeypatch.setenv("CMDIAD_DEDUP", "0")
    p_all = BatchPredictor(gpu_engine, p.bank_xyz, p.bank_second, stats, det, seg, lambdas=(1.0, 1.0, 1.0, 1.0), batch=B, workload="mtfi",
                           halluc=p.halluc, use_graph=False)
    monkeypatch.delenv("CMDIAD_DEDUP")
    img_all, pix_all = p_all.predict_batch(None, pcs)
    assert not p_all.dedup and int(p_all.live_rows.item()) == B * 3136 * p_all.xyz_searches
    assert np.array_equal(img_all, img) and np.array_equal(pix_all, pix)
    spread = float(ref_img.max() - ref_img.min())
    d_pix = np.abs(pix - ref_pix)
    print(f"mtfi [{search_dtype}] image score max |d| {np.abs(img - ref_img).max():.4f} of spread {spread:.4f}; pixel max |d| {d_pix.max():.5f} "
          f"mean {d_pix.mean():.6f} of range {np.ptp(ref_pix):.4f}")
    # both columns inherit the bf16 xyz features (the hallucinated column through the distilled MLP as well)
    assert (np.abs(img - ref_img) / np.abs(ref_img)).max() <= 0.03 and np.abs(img - ref_img).mean() <= 0.08 * spread
    # measured (round 3): mean 0.26 %, max 3.6 % of the (narrow: 0.018) map range
    assert d_pix.mean() <= 0.006 * np.ptp(ref_pix) and d_pix.max() <= 0.06 * np.ptp(ref_pix)
    i_got, i_ref = _auroc_pair(labels, img, ref_img)
    masks = np.stack([s[2].numpy().reshape(224, 224) for s in samples]).astype(int)
    p_got, p_ref = _auroc_pair(masks.ravel(), pix.ravel(), ref_pix.ravel())
    print(f"mtfi [{search_dtype}] I-AUROC {i_got:.4f} (oracle {i_ref:.4f}); P-AUROC {p_got:.4f} (oracle {p_ref:.4f})")
    assert abs(i_got - i_ref) <= 1e-2 and abs(p_got - p_ref) <= 1e-2


@pytest.mark.parametrize("tag", ["rgb", "xyz", "mtfi_xyz", "mtfi_rgb"])
def test_method_classes_vs_reference_golden(tag, golden, weights):
    """The drop-in RGBFeatures / PointFeatures / RGBorXYZWithOneHallucination(main xyz | rgb) through the five-call protocol
    against the outputs of the REFERENCE's own classes (golden G11, tests/golden/make_golden.py): statistics, normalised
    libraries (with the reference's coreset picks injected -- the greedy selection itself is pinned by G9), late-fusion rows,
    and the final image / pixel predictions under the reference's fitted one-class SVMs."""
    from cmdiad_amd.feature_extractors import multiple_features as mf
    from test_oracle_golden import g11_sample
    g = golden("g11_methods.npz")
    G = lambda k: g[f"{tag}/{k}"]  # noqa: E731
    cls, kw = {"rgb": (mf.RGBFeatures, {}), "xyz": (mf.PointFeatures, {}),
               "mtfi_xyz": (mf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="xyz")),
               "mtfi_rgb": (mf.RGBorXYZWithOneHallucination, dict(use_hn=True, main_modality="rgb"))}[tag]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = cls(make_args(f_coreset=float(g["f_coreset"]), random_state=int(g["random_state"]), **kw))
    m.deep_feature_extractor.rgb_backbone.load_state_dict(weights[0])
    m.deep_feature_extractor.xyz_backbone.load_state_dict(weights[1])
    if kw.get("use_hn"):
        m.fusion.load_state_dict(weights[2])
    sample = lambda sd, an=False: (lambda r, p: (r, p, p.clone()))(*g11_sample(g, sd, an))  # noqa: E731
    for sd in g["train_seeds"]:
        m.add_sample_to_mem_bank(sample(sd), class_name="synthetic")
    picks = [torch.from_numpy(G(f"coreset_idx{k}").astype(np.int64)) for k in range(2) if f"{tag}/coreset_idx{k}" in g.files]
    own = []
    inner = m.get_coreset_idx_randomp
    m.get_coreset_idx_randomp = lambda *a, **k: (own.append(inner(*a, **k)), picks[len(own) - 1])[1]
    m.run_coreset()
    for o, pk in zip(own, picks):  # the drop-in's own greedy selection on ITS (bf16) features: large overlap, chaotic in the last bit
        assert len(set(o.tolist()) & set(pk.tolist())) > 0.6 * len(pk), tag
    single = tag in ("rgb", "xyz")
    main_mod = tag if single else tag.split("_")[1]
    mean, std = getattr(m, f"{main_mod}_mean"), getattr(m, f"{main_mod}_std")
    np.testing.assert_allclose([float(mean), float(std)], [G("mean"), G("std")], rtol=5e-3, atol=1e-4)
    if not single:
        assert float(m.xyz_mean) == float(m.rgb_mean) == float(m.fusion_mean) and float(m.xyz_std) == float(m.rgb_std) == float(m.fusion_std)
    lib = getattr(m, f"patch_{main_mod}_lib")
    step = 31 if main_mod == "rgb" else 97
    assert lib.shape[0] == int(G("lib_rows"))
    ref_sub = G("lib_sub")
    err = np.abs(lib[::step, ::16].cpu().numpy() - ref_sub)
    assert err.mean() < 0.02 * np.abs(ref_sub).mean() and err.max() < 0.25 * np.abs(ref_sub).mean() + 0.05, (tag, err.mean(), err.max())
    if not single:
        assert m.patch_fusion_lib.shape[0] == int(G("fusion_rows"))
        ref_f = G("fusion_sub")
        err = np.abs(m.patch_fusion_lib[::97, ::16].cpu().numpy() - ref_f)
        assert err.mean() < 0.03 * np.abs(ref_f).mean() + 1e-3, (tag, err.mean())
    for sd in g["train_seeds"]:
        m.add_sample_to_late_fusion_mem_bank(sample(sd))
    s_lib = torch.cat(m.s_lib, 0).numpy()
    ref_s = G("s_lib")
    print(tag, "s_lib", s_lib.tolist(), "ref", ref_s.tolist())
    np.testing.assert_allclose(s_lib, ref_s, rtol=0.04, atol=0.02 * np.abs(ref_s).max())      # measured: <= 1.9 %
    m.run_late_fusion()
    assert m.detect_fuser.coef_.shape == G("detect_coef").shape
    # predictions under the REFERENCE's fitted models (an SGD fit on three rows flips with 1e-3 input changes)
    m.detect_fuser.coef_, m.detect_fuser.offset_ = G("detect_coef"), G("detect_offset")
    m.seg_fuser.coef_, m.seg_fuser.offset_ = G("seg_coef"), G("seg_offset")
    for sd, an in zip(g["test_seeds"], g["test_anomalous"]):
        m.predict(sample(sd, bool(an)), torch.zeros(1, 224, 224), np.array([int(an)]), ["x.png"])
    got = np.concatenate(m.image_preds).ravel()
    ref = G("image_preds")
    maps = np.stack(m.predictions)[:, ::4, ::4]
    ref_maps = G("pred_maps_sub")
    print(tag, "image_preds", got, "ref", ref, "map max |d|", np.abs(maps - ref_maps).max(), "range", np.ptp(ref_maps))
    np.testing.assert_allclose(got, ref, rtol=0.035, atol=0.01 * np.abs(ref).max())            # measured: <= 2.4 % (xyz), 0.02 % (rgb)
    assert np.abs(maps - ref_maps).mean() <= 0.015 * np.ptp(ref_maps) and np.abs(maps - ref_maps).max() <= 0.06 * np.ptp(ref_maps)   # max measured 3.7 %
    assert (got[1] > got[0]) == (ref[1] > ref[0])


def test_dropin_micro_batching_is_invisible(weights, monkeypatch):
    """The drop-in classes defer add_sample_to_mem_bank / add_sample_to_late_fusion_mem_bank / predict into micro-batches
    (CMDIAD_PREDICT_BATCH, default 16).  Whatever the batch size -- 1 = the reference's strictly-per-call behaviour -- the
    libraries, the late-fusion rows and every prediction are the same BITS, results appear in call order, and reading a
    result attribute mid-phase shows exactly the calls made so far."""
    from cmdiad_amd.feature_extractors import multiple_features as mf
    from sklearn import linear_model
    train = [synth_sample(300 + i)[:2] for i in range(3)]
    tests = [synth_sample(320 + i, anomalous=(i % 2 == 1)) for i in range(5)]
    out = {}
    for batch in ("1", "2", "8"):
        monkeypatch.setenv("CMDIAD_PREDICT_BATCH", batch)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = mf.DoubleRGBPointFeatures(make_args())
        m.deep_feature_extractor.rgb_backbone.load_state_dict(weights[0])
        m.deep_feature_extractor.xyz_backbone.load_state_dict(weights[1])
        for rgb, pc in train:
            m.add_sample_to_mem_bank((rgb, pc, pc), class_name="synthetic")
        assert len(m.patch_xyz_lib) == 3 and len(m.patch_rgb_lib) == 3          # reading flushes the pending samples
        m.run_coreset()
        for rgb, pc in train[:2]:
            m.add_sample_to_late_fusion_mem_bank((rgb, pc, pc))
        assert len(m.s_lib) == 2
        m.add_sample_to_late_fusion_mem_bank((*train[2], train[2][1]))
        s_lib = torch.cat(m.s_lib, 0).clone()
        rs = np.random.RandomState(0)
        m.detect_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 2))
        m.seg_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 2))
        for k, (rgb, pc, mask) in enumerate(tests):
            m.predict((rgb, pc, pc), mask, np.array([int(mask.any())]), [f"t{k}.png"])
            if k == 1 and batch == "2":
                # a full micro-batch is queued on the GPU but not waited for (one batch stays in flight so that its host-side
                # completion runs beside the next batch's device work): nothing recorded yet ...
                assert "predict" in m.__dict__["_inflight"] and len(m.__dict__["_lz_image_preds"]) == 0
            if k == 2:
                # ... and reading a result attribute completes it and the partial batch behind it
                assert len(m.image_preds) == 3 and [n[0] for n in m.img_name] == ["t0.png", "t1.png", "t2.png"]
                assert "predict" not in m.__dict__["_inflight"]
        m.calculate_metrics()
        assert [n[0] for n in m.img_name] == [f"t{k}.png" for k in range(5)]
        out[batch] = (m.patch_xyz_lib.cpu(), m.patch_rgb_lib.cpu(), s_lib, np.concatenate(m.image_preds).ravel(), np.stack(m.predictions),
                      float(m.image_rocauc), float(m.pixel_rocauc))
    for batch in ("2", "8"):       # per-sample results equal the B = 1 results -- bit for bit (libraries, late-fusion rows, scores, maps)
        for a, b in zip(out["1"][:3], out[batch][:3]):
            assert torch.equal(a, b)
        assert np.array_equal(out[batch][3], out["1"][3]) and np.array_equal(out[batch][4], out["1"][4])
        assert out[batch][5] == out["1"][5] and out[batch][6] == out["1"][6]


def test_bad_sample_does_not_take_its_micro_batch_with_it(weights, monkeypatch):
    """ADVICE (round 2): a sample that cannot be processed (a cloud with fewer valid points than the 128-nearest-neighbour
    grouping needs) raises -- at the flush that would have run it, INTEGRATION.md section 4 -- but the valid samples queued
    in front of it are recorded and the ones behind it stay queued: a caller that catches the error and continues loses
    exactly the bad sample, as with the reference's eager per-call loop."""
    from cmdiad_amd.feature_extractors import multiple_features as mf
    from sklearn import linear_model
    monkeypatch.setenv("CMDIAD_PREDICT_BATCH", "8")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = mf.PointFeatures(make_args())
    m.deep_feature_extractor.xyz_backbone.load_state_dict(weights[1])
    for i in range(2):
        rgb, pc, _ = synth_sample(400 + i)
        m.add_sample_to_mem_bank((rgb, pc, pc), class_name="synthetic")
    m.run_coreset()
    rs = np.random.RandomState(0)
    m.detect_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(64, 1))
    m.seg_fuser = linear_model.SGDOneClassSVM(random_state=42, nu=0.5, max_iter=1000).fit(rs.rand(4096, 1))
    good = [synth_sample(410 + i) for i in range(4)]
    bad_pc = torch.zeros(1, 3, 224, 224)
    bad_pc[0, :, 100:105, 100:110] = good[0][1][0, :, 100:105, 100:110]          # 50 valid points < group_size 128
    order = [good[0], good[1], (good[0][0], bad_pc, good[0][2]), good[2], good[3]]
    for k, (rgb, pc, mask) in enumerate(order):
        m.predict((rgb, pc, pc), mask, np.array([0]), [f"s{k}.png"])
    with pytest.raises(ValueError, match="valid points"):
        len(m.image_preds)
    assert [n[0] for n in m.__dict__["_lz_img_name"]] == ["s0.png", "s1.png"]      # the two in front were recorded
    assert [it[3][0] for it in m.__dict__["_pending"]["predict"]] == ["s3.png", "s4.png"]
    assert [n[0] for n in m.img_name] == ["s0.png", "s1.png", "s3.png", "s4.png"]  # the next read runs the ones behind it
    # and each of the four has the numbers it gets when predicted alone
    alone = []
    for rgb, pc, mask in good:
        m2_before = len(m.image_preds)
        m.predict((rgb, pc, pc), mask, np.array([0]), ["again.png"])
        alone.append(np.asarray(m.image_preds[m2_before]).ravel()[0])
    got = np.concatenate([np.asarray(v).ravel() for v in m.image_preds[:4]])
    np.testing.assert_allclose(got, np.array(alone), rtol=1e-5, atol=1e-7)
