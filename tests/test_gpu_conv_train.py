"""Hand-written training path of the convolutional FtoF head (cmdiad_amd/conv_train.py, csrc/conv_train.hip; the reference's
models/hallucination_network.py:72-147 trained by hallucination_network_pretrain.py:106-147 in train() mode) against torch
autograd in float64 on the same bf16-rounded operands: the BatchNorm + ReLU pair forward and backward, the weight gradient by
filter tap, the data gradient, a whole tower (loss, all ten gradients, running statistics).  The three-step Adam curve of the
REFERENCE's own module is tests/test_gpu_heads.py::test_head_training_follows_the_reference_loss_curve[conv_ftof] (golden G12),
which runs through this path."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cmdiad_amd import conv_train, ops  # noqa: E402

DEV = "cuda"


def _bf(t):
    return t.bfloat16().float()


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


@pytest.mark.parametrize("M,C", [(3136 * 2, 768), (1000, 64), (77, 8)])
def test_batchnorm_relu_forward_and_backward(M, C):
    g = torch.Generator().manual_seed(M + C)
    z = (torch.randn(M, C, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).double(), (0.1 * torch.randn(C, generator=g)).double()
    gamma.requires_grad_(True); beta.requires_grad_(True)
    dy = torch.randn(M, C, generator=g).double()
    y = F.relu(F.batch_norm(z, None, None, gamma, beta, training=True, eps=1e-5))
    y.backward(dy)
    zd = z.detach().float().to(DEV)
    mean64, var64 = ops.col_moments(zd)
    np.testing.assert_allclose(mean64.cpu().numpy(), z.detach().mean(0).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(var64.cpu().numpy(), z.detach().var(0, unbiased=False).numpy(), rtol=1e-5)
    rstd = (1.0 / torch.sqrt(var64 + 1e-5))
    scale = (gamma.detach().to(DEV) * rstd).float()
    shift = (beta.detach().to(DEV) - mean64 * scale.double()).float()
    got = ops.bn_relu_fwd(zd, scale, shift)
    np.testing.assert_allclose(got.float().cpu().numpy(), y.detach().numpy(), rtol=8e-3, atol=8e-3)   # bf16 output
    dz, dgamma, dbeta = ops.bn_relu_bwd(dy.float().to(DEV), zd, scale, shift, mean64.float(), rstd.float())
    np.testing.assert_allclose(dgamma.cpu().numpy(), gamma.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gamma.grad.abs().mean()))
    np.testing.assert_allclose(dbeta.cpu().numpy(), beta.grad.numpy(), rtol=2e-3, atol=2e-3 * float(beta.grad.abs().mean()))
    np.testing.assert_allclose(dz.float().cpu().numpy(), z.grad.numpy(), rtol=1e-2, atol=1e-2 * float(z.grad.abs().mean()))


@pytest.mark.parametrize("B,H,W,C,N", [(2, 56, 56, 768, 768), (3, 9, 7, 64, 72), (1, 5, 5, 128, 8), (16, 6, 6, 64, 64)])
def test_weight_and_data_gradient_of_a_3x3_convolution(B, H, W, C, N):
    g = torch.Generator().manual_seed(B + H + C + N)
    x = _bf(torch.randn(B, C, H, W, generator=g)).double()
    w = _bf(torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).double().requires_grad_(True)
    dz = _bf(torch.randn(B, N, H, W, generator=g)).double()
    x.requires_grad_(True)
    F.conv2d(x, w, None, padding=1).backward(dz)
    M = B * H * W
    x_rows = x.detach().permute(0, 2, 3, 1).reshape(M, C).contiguous().float().to(DEV).bfloat16()
    dz_rows = dz.permute(0, 2, 3, 1).reshape(M, N).contiguous().float().to(DEV).bfloat16()
    dw = conv_train._wgrad(dz_rows, x_rows, B, H, W)
    scale = float(w.grad.abs().mean())
    assert dw.shape == (N, C, 3, 3)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=2e-3, atol=2e-3 * scale)
    if N % 64 == 0:   # the data gradient is a convolution with N input channels
        dx, _ = ops.conv2d_nhwc(dz_rows.view(B, H, W, N), conv_train._conv_w_dgrad(w.detach().float().to(DEV)), C,
                                want_f32=True, want_bf16=False)
        ref = x.grad.permute(0, 2, 3, 1)
        np.testing.assert_allclose(dx.cpu().numpy(), ref.numpy(), rtol=2e-3, atol=2e-3 * float(ref.abs().mean()))


def _tower(cin, width, gen):
    layers = []
    for i in range(4):
        conv = nn.Conv2d(cin if i == 0 else width, width, 3, padding=1, bias=False)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (9 * conv.in_channels)) ** 0.5)
        layers.append(conv)
        if i < 3:
            bn = nn.BatchNorm2d(width)
            with torch.no_grad():
                bn.weight.copy_(1 + 0.2 * torch.randn(width, generator=gen)); bn.bias.copy_(0.1 * torch.randn(width, generator=gen))
            layers += [bn, nn.ReLU()]
    return nn.Sequential(*layers)


@pytest.mark.parametrize("B,side,C,sigmoid", [(2, 12, 128, False), (3, 8, 64, True), (2, 56, 768, False)])
def test_tower_loss_and_gradients_vs_torch_autograd(B, side, C, sigmoid):
    """One direction of the head: loss, the ten parameter gradients and the BatchNorm running statistics against the same
    nn.Sequential under torch autograd in float64 (CPU).  The reference rounds its forward GEMM operands to bf16 where the GPU path
    does (straight-through), so both sides open the same ReLUs -- against an un-rounded forward ~0.2 % of the ReLU masks differ,
    which alone is sqrt(0.002) = 4-5 % of gradient norm per stage (measured: cosine 0.994-0.997 after three stages).  What remains is
    the bf16 rounding of the output gradients: cosine > 0.999 per gradient tensor."""
    gen = torch.Generator().manual_seed(B * side + C)
    rt = torch.float64 if C < 768 else torch.float32      # (the full-size tower in float64 on the host would take minutes)
    ref = _tower(C, C, gen).to(rt).train()
    mine = _tower(C, C, torch.Generator().manual_seed(B * side + C)).to(DEV).train()
    T = side * side
    x, t = torch.randn(B, T, C, generator=gen), torch.randn(B, T, C, generator=gen)

    def rb(v):   # value rounded to bf16 where the GPU path rounds (GEMM operands), gradient passed straight through
        return v + (v.detach().float().bfloat16().to(v.dtype) - v.detach())

    h = rb(x.to(rt).transpose(1, 2).reshape(B, C, side, side))
    for m in ref:
        if isinstance(m, nn.Conv2d):
            h = F.conv2d(h, rb(m.weight), None, padding=1)
        elif isinstance(m, nn.BatchNorm2d):
            h = m(h)
        else:
            h = rb(F.relu(h))
    h = h.reshape(B, C, T).transpose(1, 2)
    a, b = (torch.sigmoid(h), torch.sigmoid(t.to(rt))) if sigmoid else (h, t.to(rt))
    d = torch.linalg.norm(a - b, dim=2)
    ref_loss = d.sum() / d.shape[0]
    ref_loss.backward()
    loss = conv_train.tower_loss(mine, x.to(DEV), t.to(DEV), sigmoid)
    assert loss.requires_grad
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=3e-3)
    loss.backward()
    report = {}
    for (name, p), q in zip(mine.named_parameters(), ref.parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, name
        report[name] = (_cos(p.grad.cpu(), q.grad), float((p.grad.cpu().double() - q.grad.double()).norm() / q.grad.double().norm()))
    assert all(c > 0.999 and rel < 0.05 for c, rel in report.values()), report
    for (name, u), v in zip(mine.named_buffers(), ref.buffers()):
        if u.dtype.is_floating_point:
            np.testing.assert_allclose(u.cpu().numpy(), v.double().numpy(), rtol=5e-3, atol=5e-4, err_msg=name)
        else:
            assert int(u) == int(v) == 1, name
    with torch.no_grad():   # no graph, no statistics update
        l2 = conv_train.tower_loss(mine, x.to(DEV), t.to(DEV), sigmoid)
    assert not l2.requires_grad


def test_head_module_routes_training_through_the_hip_path(monkeypatch):
    """HallucinationCrossModalityConv.forward in train() mode: the hand-written path (default) and the module's own torch layers
    (CMDIAD_CONV_TRAIN=torch, fp32 operands) give the same two losses, running statistics and -- up to the ReLU masks that bf16
    operands flip (see above: cosine 0.994-0.997 for the deepest layer) -- gradients."""
    from cmdiad_amd.models import hallucination_network as hn
    from oracle import heads
    res = {}
    gen = torch.Generator().manual_seed(9)
    a, b = torch.randn(2, 3136, 768, generator=gen), torch.randn(2, 3136, 768, generator=gen)
    for mode in ("hip", "torch"):
        monkeypatch.setenv("CMDIAD_CONV_TRAIN", mode)
        m = hn.HallucinationCrossModalityConv(None, 768, 768)
        m.load_state_dict(heads.synth_head_state_dict("conv_ftof", 41))
        m.to(DEV).train()
        lx, lr = m(a, b, False, "l2")
        (lx + lr).backward()
        res[mode] = (float(lx.detach()), float(lr.detach()), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: v.clone() for k, v in m.named_buffers()})
    np.testing.assert_allclose(res["hip"][:2], res["torch"][:2], rtol=3e-3)
    for k, gq in res["torch"][2].items():
        assert _cos(res["hip"][2][k], gq) > 0.99, k
    for k, v in res["torch"][3].items():
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(res["hip"][3][k].cpu().numpy(), v.cpu().numpy(), rtol=5e-3, atol=5e-4, err_msg=k)


@pytest.mark.parametrize("B,h,H,C", [(2, 56, 224, 384), (1, 7, 28, 8), (2, 5, 13, 4), (1, 9, 9, 4)])
def test_bicubic_upsampling_adjoint_vs_torch(B, h, H, C):
    """cmdiad_upsample_bicubic_bwd against autograd through F.interpolate(mode='bicubic', align_corners=False) in float64:
    the x4 case of the heads, other ratios, border clamping, identity size."""
    g = torch.Generator().manual_seed(B + h + H + C)
    x = torch.randn(B, C, h, h, generator=g).double().requires_grad_(True)
    go = torch.randn(B, C, H, H, generator=g)
    F.interpolate(x, size=(H, H), mode="bicubic", align_corners=False).backward(go.double())
    got = ops.upsample_bicubic_bwd(go.permute(0, 2, 3, 1).contiguous().to(DEV), h, h)
    ref = x.grad.permute(0, 2, 3, 1)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().mean()))


def test_relu_backward_from_the_saved_output():
    g = torch.Generator().manual_seed(4)
    y = torch.relu(torch.randn(1000, 64, generator=g)).bfloat16()
    dx = torch.randn(1000, 64, generator=g)
    got = ops.relu_bwd(dx.to(DEV), y.to(DEV))
    want = torch.where(y.float() > 0, dx, torch.zeros_like(dx)).bfloat16()
    assert torch.equal(got.cpu(), want)


def test_feature_to_input_conv_head_trains_on_the_hip_path(monkeypatch):
    """HallucinationFeatureToInputConv.forward in train() mode: hand-written path (default) against the module's own torch layers
    (CMDIAD_CONV_TRAIN=torch, fp32): loss 3e-3, gradient cosines (ReLU masks flip under bf16 operands: > 0.99)."""
    from cmdiad_amd.models import hallucination_network as hn
    from oracle import heads
    gen = torch.Generator().manual_seed(12)
    f, img = torch.randn(2, 3136, 768, generator=gen), torch.randn(2, 3, 224, 224, generator=gen)
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("CMDIAD_CONV_TRAIN", mode)
        m = hn.HallucinationFeatureToInputConv(None, 768)
        m.load_state_dict(heads.synth_head_state_dict("ftoi_conv", 41))
        m.to(DEV).train()
        loss = m(f, img)
        assert loss.requires_grad
        loss.backward()
        res[mode] = (float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    np.testing.assert_allclose(res["hip"][0], res["torch"][0], rtol=3e-3)
    assert set(res["hip"][1]) == set(res["torch"][1]) == {f"conv{i}.{w}" for i in range(1, 5) for w in ("weight", "bias")}
    for k, gq in res["torch"][1].items():
        assert res["hip"][1][k].shape == gq.shape and _cos(res["hip"][1][k], gq) > 0.99, (k, _cos(res["hip"][1][k], gq))


@pytest.mark.parametrize("depth_only", [False, True])
def test_feature_to_input_mlp_head_trains_on_the_hip_path(depth_only, monkeypatch):
    """HallucinationRGBFeatureToXYZInputMLP.forward in train() mode (3 output channels, and 1 with --estimate_depth): hand-written
    path against the module's own torch layers (fp32): loss 3e-3, every gradient cosine > 0.995 (GELU is smooth: no mask flips)."""
    import types
    from cmdiad_amd.models import hallucination_network as hn
    from oracle import heads
    gen = torch.Generator().manual_seed(13)
    f = torch.randn(2, 3136, 768, generator=gen)
    img = torch.randn(2, 1 if depth_only else 3, 224, 224, generator=gen)
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("CMDIAD_CONV_TRAIN", mode)
        torch.manual_seed(5)
        m = hn.HallucinationRGBFeatureToXYZInputMLP(types.SimpleNamespace(estimate_depth=depth_only), 768)
        if not depth_only:
            m.load_state_dict(heads.synth_head_state_dict("ftoi_mlp", 41))
        m.to(DEV).train()
        loss = m(f, img)
        assert loss.requires_grad
        loss.backward()
        res[mode] = (float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters()})
    np.testing.assert_allclose(res["hip"][0], res["torch"][0], rtol=3e-3)
    for k, gq in res["torch"][1].items():
        c = _cos(res["hip"][1][k], gq)
        assert res["hip"][1][k].shape == gq.shape and c > 0.995, (k, c)


def test_hrnet_trunk_trains_on_the_hip_path(monkeypatch):
    """models.hrnet.HRNet.forward in train() mode (stem, twelve Bottlenecks, final 1x1; 39 batch-statistics BatchNorms): hand-written
    path against the module's own torch layers (fp32): loss 3e-3, running statistics, and the gradient of every parameter the
    forward uses -- cosine > 0.999 at the final layer, > 0.98 in the last Bottleneck, falling with depth as bf16 operands flip
    ~0.2 % of every ReLU mask on the way down (36 ReLUs between the loss and the stem: 1 - 36 x 0.002 / 2 = 0.964; measured 0.96-0.97
    at the stem)."""
    from cmdiad_amd.models.hrnet import HRNet
    from oracle import heads
    gen = torch.Generator().manual_seed(21)
    img, feat = torch.randn(2, 3, 224, 224, generator=gen), torch.randn(2, 3136, 768, generator=gen)
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("CMDIAD_HRNET_TRAIN", mode)      # (auto = hip, see hrnet.py)
        m = HRNet(512, 768, 0.1)
        m.load_state_dict(heads.synth_head_state_dict("hrnet", 41))
        m.to(DEV).train()
        loss = m(img, feat)
        assert loss.requires_grad
        loss.backward()
        res[mode] = (float(loss.detach()), {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()},
                     {k: v.clone() for k, v in m.named_buffers()})
    np.testing.assert_allclose(res["hip"][0], res["torch"][0], rtol=3e-3)
    cos = {}
    for k, gq in res["torch"][1].items():
        gh = res["hip"][1][k]
        assert (gh is None) == (gq is None), k                      # layer4 is constructed but never run: no gradient on either side
        if gq is not None:
            assert gh.shape == gq.shape, k
            cos[k] = _cos(gh, gq)
    assert cos["final_layer.weight"] > 0.999 and cos["final_layer.bias"] > 0.999, cos
    assert min(v for k, v in cos.items() if k.startswith("layer3.3")) > 0.98, cos
    assert min(cos.values()) > 0.93, cos
    for k, v in res["torch"][2].items():
        if v.dtype.is_floating_point and not k.startswith("layer4."):
            np.testing.assert_allclose(res["hip"][2][k].cpu().numpy(), v.cpu().numpy(), rtol=1e-2, atol=2e-3, err_msg=k)


def test_hrnet_step_replayed_as_a_graph_equals_the_eager_step(monkeypatch):
    """conv_train._hrnet_step: from the third step with the same shapes and parameter storages the trunk's forward + backward is ONE
    HIP-graph replay.  Five SGD steps on two copies of the module -- graph replay on / off -- give the same losses, parameters and
    running statistics bit for bit (the graph holds the very launches of the eager step); the loss tensor of an earlier step keeps its
    value; and a backward() whose gradients a later forward has overwritten raises instead of using them."""
    from cmdiad_amd.models.hrnet import HRNet
    from oracle import heads
    gen = torch.Generator().manual_seed(22)
    batches = [(torch.randn(2, 3, 224, 224, generator=gen).to(DEV), torch.randn(2, 3136, 768, generator=gen).to(DEV)) for _ in range(5)]
    monkeypatch.setenv("CMDIAD_HRNET_TRAIN", "hip")
    out = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("CMDIAD_HRNET_GRAPH", graph)
        conv_train._HRNET_GRAPHS.clear()
        m = HRNet(512, 768, 0.1)
        m.load_state_dict(heads.synth_head_state_dict("hrnet", 41))
        m.to(DEV).train()
        opt = torch.optim.SGD(m.parameters(), lr=1e-3)
        losses = []
        for img, feat in batches:
            opt.zero_grad()
            loss = m(img, feat)
            loss.backward()
            opt.step()
            losses.append(loss)                      # kept as tensors: a replay must not change an earlier step's loss
        if graph == "1":
            assert any("graph" in e for e in conv_train._HRNET_GRAPHS.values())        # steps 3-5 were replays
        out[graph] = ([float(l.detach()) for l in losses], {k: v.detach().clone() for k, v in m.state_dict().items()})
    assert out["1"][0] == out["0"][0], (out["1"][0], out["0"][0])
    assert len(set(out["1"][0])) == 5
    for k, v in out["0"][1].items():
        assert torch.equal(out["1"][1][k], v), k
    # two forwards, then the first one's backward: its gradients are gone
    monkeypatch.setenv("CMDIAD_HRNET_GRAPH", "1")
    conv_train._HRNET_GRAPHS.clear()
    m = HRNet(512, 768, 0.1)
    m.load_state_dict(heads.synth_head_state_dict("hrnet", 41))
    m.to(DEV).train()
    for img, feat in batches[:3]:
        m(img, feat).backward()
    first = m(*batches[3])
    m(*batches[4])
    with pytest.raises(RuntimeError, match="overwritten"):
        first.backward()
