"""GPU parity of the convolution kernels behind the conv / FtoI / HRNet distillation heads:
cmdiad_conv2d_nhwc_bf16 (implicit GEMM), cmdiad_conv_stem, cmdiad_upsample_bicubic, each against the torch fp32 op the
reference calls (nn.Conv2d / F.interpolate(mode='bicubic'); models/hallucination_network.py:72-220, models/hrnet.py).

Tolerance: operands are rounded to bf16 for the MFMA product (inputs here are pre-rounded to bf16 so only the
accumulation order differs: fp32 accumulate, |err| <= 2e-3 of the output scale); bicubic / stem are fp32 arithmetic with
a different summation order (1e-5 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cmdiad_amd import ops  # noqa: E402

DEV = "cuda"


def _bf(t):
    return t.to(torch.bfloat16).float()


def _pack(w):  # torch [N,C,kh,kw] -> [N, kh*kw*C] bf16 (tap-major)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16)


@pytest.mark.parametrize("B,H,W,C,N,ks,stride,act,res", [
    (2, 56, 56, 128, 128, 3, 1, "relu", False),      # Bottleneck conv2 (hrnet.py:15)
    (1, 56, 56, 768, 768, 3, 1, "relu", False),      # HallucinationCrossModalityConv layer
    (2, 30, 22, 64, 132, 3, 2, "none", False),       # stride 2, ragged sizes, N not a tile multiple
    (1, 112, 112, 64, 128, 3, 2, "relu", False),     # HRNet stem conv2 (hrnet.py:152)
    (3, 17, 9, 192, 68, 3, 1, "relu_post", True),    # residual + ReLU after it, M not a tile multiple
    (2, 14, 14, 512, 128, 1, 1, "relu", False),      # 1x1 (Bottleneck conv1)
    (2, 14, 14, 128, 512, 1, 1, "relu_post", True),  # 1x1 + residual (Bottleneck conv3)
    (1, 9, 9, 64, 4, 3, 1, "none", False),           # tiny N (the FtoI head's last layer, 3 -> padded 4)
])
def test_conv2d_nhwc_vs_torch(B, H, W, C, N, ks, stride, act, res):
    g = torch.Generator().manual_seed(B * 1000 + C + N)
    x = _bf(torch.randn(B, C, H, W, generator=g))
    w = _bf(torch.randn(N, C, ks, ks, generator=g) / (C * ks * ks) ** 0.5)
    bias = torch.randn(N, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=1 if ks == 3 else 0)
    residual = torch.randn(ref.shape, generator=g) if res else None
    if act == "relu":
        ref = ref.relu()
    if res:
        ref = ref + residual.double()
    if act == "relu_post":
        ref = ref.relu()
    ref = ref.permute(0, 2, 3, 1).float()  # NHWC
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    r = residual.permute(0, 2, 3, 1).contiguous().to(DEV) if res else None
    o32, o16 = ops.conv2d_nhwc(x_nhwc, _pack(w).to(DEV), N, ks, stride, bias=bias.to(DEV),
                               act={"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "relu_post": ops.ACT_RELU_POST}[act],
                               residual=r, want_f32=True, want_bf16=True)
    assert o32.shape == ref.shape
    scale = ref.abs().mean().item()
    assert (o32.cpu() - ref).abs().max().item() <= 2e-3 * max(scale, 1.0)
    assert (o16.float().cpu() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1.0)


def test_conv2d_padded_output_columns_stay_untouched():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 12, 12, 64, generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24
    out = torch.full((1, 12, 12, 128), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.conv2d_nhwc(x, _pack(w).to(DEV), 96, 3, 1, out_bf16=out, want_bf16=False)
    assert (out[..., 96:] == 7.0).all() and not (out[..., :96] == 7.0).all()


def test_conv2d_rejects_bad_arguments():
    from cmdiad_amd._native import NativeError
    x = torch.zeros(1, 8, 8, 48, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 9 * 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(NativeError, match="C%64"):
        ops.conv2d_nhwc(x, w, 64)
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(NativeError, match="3x3"):
        ops.conv2d_nhwc(x, torch.zeros(64, 25 * 64, dtype=torch.bfloat16, device=DEV), 64, ksize=5)


@pytest.mark.parametrize("stride,Cin,Cout,H,W", [(2, 3, 64, 224, 224), (1, 1, 8, 13, 7), (2, 4, 40, 31, 18)])
def test_conv_stem_vs_torch(stride, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(Cout)
    x = torch.randn(2, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=1).relu().permute(0, 2, 3, 1)
    got = ops.conv_stem(x.to(DEV), w.to(DEV), b.to(DEV), stride).float().cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-5  # one bf16 rounding of the output


@pytest.mark.parametrize("B,h,w,C,H,W", [(2, 56, 56, 384, 224, 224), (1, 56, 56, 3, 224, 224), (2, 7, 5, 6, 19, 23), (1, 56, 56, 1, 224, 224),
                                         (3, 6, 5, 8, 24, 20), (1, 2, 1, 4, 8, 4)])  # exact 4x: the block kernel, borders everywhere
def test_bicubic_vs_torch(B, h, w, C, H, W):
    g = torch.Generator().manual_seed(C)
    ld = (C + 3) // 4 * 4
    x = torch.randn(B, h, w, ld, generator=g)
    ref = F.interpolate(x[..., :C].permute(0, 3, 1, 2).contiguous(), size=(H, W), mode="bicubic")
    got = ops.upsample_bicubic(x.to(DEV), C, H, W, nchw=True).cpu()
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-6)
    out = torch.zeros(B, H, W, ld + 4, dtype=torch.bfloat16, device=DEV)
    ops.upsample_bicubic(x.to(DEV), C, H, W, out_bf16=out)
    assert (out[..., C:] == 0).all()
    assert (out[..., :C].float().cpu() - ref.permute(0, 2, 3, 1)).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6
