"""GPU: training path of the FtoF distillation net (losses, gradients, Adam steps) against the reference's
golden vectors (G5, produced by the reference's own module + torch.optim.Adam) and a torch fp32 reference."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import train  # noqa: E402
from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork  # noqa: E402
from cmdiad_amd.utils import lr_sched  # noqa: E402
from oracle import nets  # noqa: E402

DEV = "cuda"


def _net():
    net = HallucinationCrossModalityNetwork(None, 768, 768, hidden_ratio=2.5, mlp_depth=1)
    net.load_state_dict(nets.synth_state_dict("halluc", 51), strict=True)
    return net.to(DEV)


def _samples(g):
    s = torch.randn(2, 64, 1536, generator=torch.Generator().manual_seed(int(g["samples_seed"])))
    return s[:, :, :768].contiguous(), s[:, :, 768:].contiguous()


def test_state_dict_keys_match_reference_names():
    keys = set(HallucinationCrossModalityNetwork(None, 768, 768).state_dict())
    assert keys == set(nets.synth_state_dict("halluc", 51))
    assert sum(p.numel() for p in HallucinationCrossModalityNetwork(None, 768, 768).parameters()) == 13283328


def test_losses_match_golden(golden):
    g = golden("g5_halluc.npz")
    net = _net()
    xyz, rgb = _samples(g)
    with torch.no_grad():
        for dm in ("l2", "cos_dist", "smooth_l1"):
            a, b = net(xyz, rgb, False, dm)
            # bf16 GEMM chain: outputs carry ~4e-3 absolute error, summed over 64 x 768 elements per sample
            np.testing.assert_allclose([a.item(), b.item()], g[f"loss_{dm}"], rtol=1e-2)


def test_gradients_match_torch_autograd():
    sd = nets.synth_state_dict("halluc", 51)
    s = torch.randn(2, 96, 1536, generator=torch.Generator().manual_seed(3))
    xyz, rgb = s[:, :, :768].contiguous(), s[:, :, 768:].contiguous()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    for dm in ("l2", "cos_dist", "smooth_l1"):
        lx, lr = nets.halluc_losses(params, xyz, rgb, dm)
        ref = torch.autograd.grad(lx + lr, list(params.values()))
        net = _net()
        ax, ar = net(xyz, rgb, False, dm)
        (ax + ar).backward()
        got = dict(net.named_parameters())
        for (name, _), r in zip(params.items(), ref):
            gq = got[name].grad.cpu()
            # cosine similarity + norm ratio: gradients flow through three bf16 GEMMs each way
            cos = torch.nn.functional.cosine_similarity(gq.flatten(), r.flatten(), dim=0).item()
            ratio = (gq.norm() / r.norm()).item()
            assert cos > 0.995 and abs(ratio - 1) < 0.02, (dm, name, cos, ratio)


def test_mlp_depth_2_matches_reference_golden_and_autograd(golden):
    """mlp_depth = 2 (utils/utils.py:103-115, hallucination_network_pretrain.py:70,247): generated features and the three
    losses against the reference's own module (golden G5b), the loss curve of three warm-up Adam steps, and every
    parameter gradient against torch fp32 autograd through the oracle restatement."""
    import types
    g = golden("g5b_halluc_depth2.npz")
    sd = nets.synth_state_dict("halluc", int(g["weights_seed"]), mlp_depth=2)
    net = HallucinationCrossModalityNetwork(None, 768, 768, hidden_ratio=2.5, mlp_depth=2)
    assert set(net.state_dict()) == set(sd)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV)
    s = torch.randn(2, 32, 1536, generator=torch.Generator().manual_seed(int(g["samples_seed"])))
    xyz, rgb = s[:, :, :768].contiguous(), s[:, :, 768:].contiguous()
    with torch.no_grad():
        for got, key in ((net.hallucination_generation(xyz_feature=xyz, out_type="rgb"), "gen_xyz2rgb"),
                         (net.hallucination_generation(rgb_feature=rgb, out_type="xyz"), "gen_rgb2xyz")):
            err = (got.cpu().numpy() - g[key])
            assert np.abs(err).mean() < 4e-3 and np.abs(err).max() < 5e-2, (key, np.abs(err).mean(), np.abs(err).max())
        for dm in ("l2", "cos_dist", "smooth_l1"):
            a, b = net(xyz, rgb, False, dm)
            np.testing.assert_allclose([a.item(), b.item()], g[f"loss_{dm}"], rtol=1e-2)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lx, lr_ = nets.halluc_losses(params, xyz, rgb, "l2")
    ref = torch.autograd.grad(lx + lr_, list(params.values()))
    ax, ar = net(xyz, rgb, False, "l2")
    (ax + ar).backward()
    got = dict(net.named_parameters())
    for (name, _), r in zip(params.items(), ref):
        gq = got[name].grad.cpu()
        cos = torch.nn.functional.cosine_similarity(gq.flatten(), r.flatten(), dim=0).item()
        ratio = (gq.norm() / r.norm()).item()
        assert cos > 0.99 and abs(ratio - 1) < 0.03, (name, cos, ratio)      # six bf16 GEMMs each way
    net.zero_grad()
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    sargs = types.SimpleNamespace(lr=5e-4, warmup_epochs=1, epochs=10)
    net.train()
    for it in range(3):
        lr_sched.adjust_learning_rate(opt, it / 4 + 0, sargs)
        lx, lr_ = net(xyz, rgb, False, "l2")
        np.testing.assert_allclose([lx.item(), lr_.item()], g["train_losses"][it], rtol=1e-2)
        (lx + lr_).backward()
        opt.step()
        opt.zero_grad()


@pytest.mark.parametrize("opt_kind", ["torch", "fused"])
def test_adam_steps_match_golden(golden, opt_kind):
    """Three update steps exactly as hallucination_network_pretrain.py:102-154 drives them."""
    import types
    g = golden("g5_halluc.npz")
    net = _net()
    xyz, rgb = _samples(g)
    opt = torch.optim.Adam(net.parameters(), lr=5e-4) if opt_kind == "torch" else train.FusedAdam(net.parameters(), lr=5e-4)
    sargs = types.SimpleNamespace(lr=5e-4, warmup_epochs=1, epochs=10)
    net.train()
    opt.zero_grad()
    for it in range(3):
        lr_sched.adjust_learning_rate(opt, it / 4 + 0, sargs)
        lx, lr_ = net(xyz, rgb, False, "l2")
        np.testing.assert_allclose([lx.item(), lr_.item()], g["train_losses"][it], rtol=1e-2)
        (lx + lr_).backward()
        opt.step()
        opt.zero_grad()
        if it in (0, 2):
            named = dict(net.named_parameters())
            for pn in ("xyz_mlp.mlp_module.0.fc1.weight", "rgb_mlp.mlp_module.0.fc3.bias", "xyz_norm.weight"):
                t = named[pn].detach().cpu()
                got = (t[:8, :8] if t.dim() == 2 else t[:16]).numpy()
                ref = g[f"step{it + 1}_{pn}"]
                # Adam moves every weight by ~lr per step whatever the gradient scale: compare the UPDATE
                init = nets.synth_state_dict("halluc", 51)[pn]
                init = (init[:8, :8] if init.dim() == 2 else init[:16]).numpy()
                du, dr = got - init, ref - init
                if np.abs(dr).max() > 0:
                    assert np.abs(du - dr).max() <= 0.35 * np.abs(dr).max() + 1e-7, (pn, it, du, dr)
                    assert np.sign(du[np.abs(dr) > 0.5 * np.abs(dr).max()]).tolist() == np.sign(dr[np.abs(dr) > 0.5 * np.abs(dr).max()]).tolist()


def test_config3_step_full_size_loss_vs_oracle():
    """BASELINE configs[2] shape: batch 32 x 3136 tokens, one full step (forward, backward, Adam).  The loss is a sum over
    tokens divided by the batch size (hallucination_network.py:47-69), so the full-size value is checked through that
    linearity -- the four quarter-batches' losses add up to it -- and anchored to the fp32 oracle on a two-sample subset
    (2 x 3136 tokens, both directions); the step then moves every parameter by about lr."""
    net = _net()
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    s = torch.randn(32, 3136, 1536, generator=torch.Generator().manual_seed(3407)).to(DEV)
    with torch.no_grad():
        parts = [net(s[i:i + 8, :, :768], s[i:i + 8, :, 768:], False, "l2") for i in range(0, 32, 8)]
        sub = net(s[[3, 20], :, :768], s[[3, 20], :, 768:], False, "l2")
    sd = nets.synth_state_dict("halluc", 51)
    with torch.no_grad():
        ox, orr = nets.halluc_losses(sd, s[[3, 20], :, :768].cpu(), s[[3, 20], :, 768:].cpu(), "l2")
    np.testing.assert_allclose([sub[0].item(), sub[1].item()], [ox.item(), orr.item()], rtol=5e-3)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    lx, lr_ = net(s[:, :, :768], s[:, :, 768:], False, "l2")
    np.testing.assert_allclose(lx.item(), sum(p[0].item() for p in parts) * 8 / 32, rtol=2e-4)
    np.testing.assert_allclose(lr_.item(), sum(p[1].item() for p in parts) * 8 / 32, rtol=2e-4)
    (lx + lr_).backward()
    opt.step()
    torch.cuda.synchronize()
    for k, v in net.named_parameters():
        d = (v.detach() - before[k]).abs()
        assert torch.isfinite(v).all() and 1e-4 < float(d.max()) <= 5e-4 * 1.001, (k, float(d.max()))   # Adam's first step: |update| <= lr


def test_feature_ring_on_device(tmp_path):
    """FeatureRing with pinned staging + copy stream: every batch equals the files it was built from, also when the
    consumer lags (ring slots are recycled only behind the consumer's queued work)."""
    from cmdiad_amd.dataset import FeatureRing, PreTrainTensorDataset
    g = torch.Generator().manual_seed(9)
    for i in range(13):
        torch.save(torch.randn(64, 1536, generator=g), tmp_path / f"cable{i}.pt")
    ds = PreTrainTensorDataset(str(tmp_path))
    t, lab = ds[3]
    assert t.is_cuda and t.shape == (64, 1536) and lab == 0
    torch.manual_seed(1)
    ring = FeatureRing(str(tmp_path), 4, shuffle=True, drop_last=True, device="cuda", depth=3, readers=3)
    torch.manual_seed(1)
    plan = ring.batches()
    torch.manual_seed(1)
    sums = []
    for x, lab in ring:
        assert x.is_cuda and x.shape == (4, 64, 1536)
        y = x
        for _ in range(20):  # keep the stream busy so the refill has to wait for this batch's consumers
            y = y * 1.0000001
        sums.append((x.double().sum(dim=(1, 2)), y))
    assert len(sums) == len(plan) == 3
    files = os.listdir(tmp_path)
    for (s, _), idxs in zip(sums, plan):
        want = torch.stack([torch.load(tmp_path / files[i]).double().sum() for i in idxs])
        np.testing.assert_allclose(s.cpu().numpy(), want.numpy(), rtol=1e-12)
    # later epochs are served from the HBM-resident cache (samples seen so far) or from disk (the rest): same contents
    for epoch in range(3):
        torch.manual_seed(10 + epoch)
        plan = ring.batches()
        torch.manual_seed(10 + epoch)
        for (x, _), idxs in zip(ring, plan):
            want = torch.stack([torch.load(tmp_path / files[i]) for i in idxs])
            assert torch.equal(x.cpu(), want)
    assert sum(ring._cached) >= 12


@pytest.mark.parametrize("M,N1,N2,split", [(512, 768, 1920, 1), (4096, 1920, 1920, 8), (1024, 136, 72, 3), (64, 8, 8, 1),
                                           (640, 200, 328, 16)])
def test_gemm_tn_vs_torch(M, N1, N2, split):
    """cmdiad_gemm_tn_bf16 (the weight-gradient product with both operands row-major, transposing LDS reads) against
    P^T Q in float64 on the bf16-rounded operands: fp32 accumulation, error <= 1e-3 of the output scale.  Ragged N (not a
    tile multiple), an uneven / over-long split (empty slabs must be written as zeros), the smallest legal shape."""
    from cmdiad_amd import ops
    g = torch.Generator().manual_seed(M + N1)
    P = torch.randn(M, N1, generator=g).to(torch.bfloat16)
    Q = (torch.randn(M, N2, generator=g) + 0.1 * torch.arange(N2)[None] / N2).to(torch.bfloat16)  # asymmetric columns
    ref = P.double().T @ Q.double()
    if split > M // 64:
        split = M // 64
    out, cs = ops.gemm_tn(P.cuda(), Q.cuda(), split_k=split, want_colsum=True)
    got = (out.sum(0) if split > 1 else out).double().cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-4
    cs = (cs.sum(0) if split > 1 else cs).double().cpu()   # the bias gradient from the same staged tiles
    ref_cs = P.double().sum(0)
    assert cs.shape == ref_cs.shape and (cs - ref_cs).abs().max().item() <= 1e-3 * ref_cs.abs().max().item() + 1e-3
    assert torch.equal(ops.gemm_tn(P.cuda(), Q.cuda(), split_k=split), out)


@pytest.mark.parametrize("dm", ["l2", "cos_dist"])
def test_loss_curve_matches_fp32_reference_over_30_steps(dm):
    """hallucination_network_pretrain.py:102-159 for 30 update steps (linear warm-up then constant rate, lr_sched.py:4-17,
    Adam with default betas, no weight decay) on a fixed batch: the HIP trainer (bf16 GEMM operands, fp32 accumulate, fp32
    master weights, fused Adam) against the same loop in torch fp32 on the CPU (oracle.nets.halluc_losses + torch.optim.Adam,
    the arithmetic golden G5 pins for the first three steps).  Every step's two losses within 1 %, the accumulated
    parameter change of every tensor pointing the same way (cosine), equal length within 3 %."""
    import types
    sd = nets.synth_state_dict("halluc", 51)
    s = torch.randn(4, 128, 1536, generator=torch.Generator().manual_seed(11))
    xyz, rgb = s[:, :, :768].contiguous(), s[:, :, 768:].contiguous()
    sargs = types.SimpleNamespace(lr=5e-4, warmup_epochs=1, epochs=10)
    steps, per_epoch = 30, 10
    # ---- torch fp32 reference
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ropt = torch.optim.Adam(list(params.values()), lr=5e-4)
    ref_losses = []
    for it in range(steps):
        lr_sched.adjust_learning_rate(ropt, it / per_epoch, sargs)
        lx, lr_ = nets.halluc_losses(params, xyz, rgb, dm)
        ref_losses.append([lx.item(), lr_.item()])
        ropt.zero_grad()
        (lx + lr_).backward()
        ropt.step()
    # ---- HIP trainer
    net = _net()
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    net.train()
    got_losses = []
    for it in range(steps):
        lr_sched.adjust_learning_rate(opt, it / per_epoch, sargs)
        lx, lr_ = net(xyz, rgb, False, dm)
        got_losses.append([lx.item(), lr_.item()])
        opt.zero_grad()
        (lx + lr_).backward()
        opt.step()
    ref_losses, got_losses = np.array(ref_losses), np.array(got_losses)
    assert ref_losses[-1].sum() < 0.9 * ref_losses[0].sum()                      # the loop does learn on this batch
    np.testing.assert_allclose(got_losses, ref_losses, rtol=1e-2)
    named = dict(net.named_parameters())
    for k, p0 in sd.items():
        du = (named[k].detach().cpu() - p0).flatten().double()
        dr = (params[k].detach() - p0).flatten().double()
        cos = float(du @ dr / (du.norm() * dr.norm() + 1e-30))
        assert cos > 0.98 and abs(float(du.norm() / dr.norm()) - 1) < 0.03, (k, cos, float(du.norm() / dr.norm()))
