"""LayerNorm folded into the products on either side of it (csrc/gemm.hip "LayerNorm fold", ABI 3; the reference's
Block.forward, models/models.py:177-180: x = x + attn(norm1(x)); x = x + mlp(norm2(x))).

Producer: the in-place residual product also emits the new rows as bf16 and per-64-column (sum, M2) partials;
cmdiad_ln_stats_finalize merges them into 1 / sigma; consumer: row_scale in cmdiad_gemm_bf16 (128 x 128 and the persistent
256 x 256 kernel) and cmdiad_gemm_qkv.  Checked against torch in float64 at the kernel level, against the separate
LayerNorm launches (CMDIAD_LN_FOLD=0) and the fp32 oracle at the network level; the one-call block entry point must equal
the same launches issued one by one, bit for bit, with the fold chained across blocks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cmdiad_amd import ops, runtime  # noqa: E402
from oracle import nets  # noqa: E402

DEV = "cuda"


def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("M,N,K,with_add", [(256, 128, 64, False), (1000, 384, 384, True), (785 * 2, 768, 768, False),
                                             (3 * 1024 + 5, 384, 1536, True), (77, 64, 192, True)])
def test_residual_product_emits_rows_and_statistics(M, N, K, with_add):
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias, res = torch.randn(N, generator=g), 3.0 * torch.randn(M, N, generator=g) + 0.7
    add = 0.2 * torch.randn(M, N, generator=g) if with_add else None
    dA, dW = A.to(DEV).bfloat16(), W.to(DEV).bfloat16()
    # the plain in-place form is the yardstick: the fold's fp32 rows must equal it bit for bit (without the second addend)
    x0 = res.clone().to(DEV)
    ops.gemm(dA, dW, bias=bias.to(DEV), residual=x0, out_f32=x0, want_bf16=False)
    x = res.clone().to(DEV)
    xb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    part = torch.full((N // 64, M, 2), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(dA, dW, bias=bias.to(DEV), residual=x, out_f32=x, want_bf16=False, ln_xb=xb, ln_part=part,
             add2=add.to(DEV) if with_add else None)
    want = x0 + add.to(DEV) if with_add else x0
    assert torch.equal(x, want)                                  # (x + f(x)) + pos, in that order
    assert torch.equal(xb, x.bfloat16())                         # the same rows, rounded once
    xd = x.double().cpu().reshape(M, N // 64, 64)
    s_ref = xd.sum(-1).T                                         # [chunks, M]
    q_ref = ((xd - xd.mean(-1, keepdim=True)) ** 2).sum(-1).T
    np.testing.assert_allclose(part[..., 0].cpu().numpy(), s_ref.numpy(), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(part[..., 1].cpu().numpy(), q_ref.numpy(), rtol=2e-5, atol=1e-5)
    for eps in (1e-5, 1e-6):
        rstd, mean = ops.ln_stats_finalize(part, M, N // 64, eps, want_mean=True)
        assert rstd.numel() == (M + 255) // 256 * 256
        var = x.double().cpu().var(dim=1, unbiased=False)
        np.testing.assert_allclose(rstd[:M].cpu().numpy(), (1.0 / torch.sqrt(var + eps)).numpy(), rtol=3e-6)
        np.testing.assert_allclose(mean.cpu().numpy(), x.double().cpu().mean(1).numpy(), rtol=1e-5, atol=1e-6)


def test_statistics_of_rows_with_a_large_common_offset():
    """Chunk partials are deviations from the CHUNK mean and are merged with Chan's formula: no E[x^2] - mean^2 cancellation
    when a row's mean is 1000 standard deviations."""
    M, N, K = 300, 256, 64
    g = torch.Generator().manual_seed(5)
    A = torch.zeros(M, K)
    W = torch.zeros(N, K)
    res = 1000.0 + torch.randn(M, N, generator=g)
    x = res.clone().to(DEV)
    xb = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    part = torch.empty((N // 64, M, 2), dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), bias=torch.zeros(N, device=DEV), residual=x, out_f32=x, want_bf16=False,
             ln_xb=xb, ln_part=part)
    rstd = ops.ln_stats_finalize(part, M, N // 64, 1e-6)
    var = res.double().var(dim=1, unbiased=False)
    np.testing.assert_allclose(rstd[:M].cpu().numpy(), (1.0 / torch.sqrt(var + 1e-6)).numpy(), rtol=1e-4)


@pytest.mark.parametrize("M,N,K,pp3", [(1000, 512, 192, "0"), (1000, 512, 192, "1"), (3 * 785, 1536, 768, "1"), (3 * 785, 1536, 768, "0"),
                                       (40000, 512, 256, "1"), (130, 100, 64, "0")])
def test_row_scale_epilogue(M, N, K, pp3, monkeypatch):
    """out = act(row_scale[m] * acc + bias) on the 128 x 128 kernel and on the persistent 256 x 256 kernel (identical bits),
    ragged last tiles included; row_scale is allocated to M rounded up to 256 as the ABI asks."""
    monkeypatch.setenv("CMDIAD_GEMM_PP3", pp3)
    g = torch.Generator().manual_seed(M + N)
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    rs = torch.full(((M + 255) // 256 * 256,), float("nan"))
    rs[:M] = 0.5 + torch.rand(M, generator=g)
    dA, dW, drs = A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), rs.to(DEV)
    ref = (A.double() @ W.double().T) * rs[:M, None].double() + bias.double()
    for act, f in ((ops.ACT_GELU, torch.nn.functional.gelu), (ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, torch.relu)):
        _, o16 = ops.gemm(dA, dW, bias=bias.to(DEV), act=act, row_scale=drs)
        assert torch.isfinite(o16.float()).all()
        np.testing.assert_allclose(o16.float().cpu().numpy(), f(ref).numpy(), rtol=8e-3, atol=8e-3)
        monkeypatch.setenv("CMDIAD_GEMM_PP3", "0")
        _, base = ops.gemm(dA, dW, bias=bias.to(DEV), act=act, row_scale=drs)
        monkeypatch.setenv("CMDIAD_GEMM_PP3", pp3)
        assert torch.equal(o16, base)
    o32, _ = ops.gemm(dA, dW, bias=bias.to(DEV), row_scale=drs, want_f32=True, want_bf16=False)
    np.testing.assert_allclose(o32.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,T,C", [(2, 785, 768), (1, 1024, 384), (3, 200, 128)])
def test_qkv_row_scale(B, T, C):
    H, Tp, M = C // 64, (T + 63) // 64 * 64, B * T
    g = torch.Generator().manual_seed(B * T + C)
    x = _bf(torch.randn(M, C, generator=g))
    W = _bf(torch.randn(3 * C, C, generator=g) / C ** 0.5)
    bias = 0.1 * torch.randn(3 * C, generator=g)
    rs = 0.5 + torch.rand(M, generator=g)
    q = torch.zeros(B, H, Tp, 64, dtype=torch.bfloat16, device=DEV)
    k, vt = torch.zeros_like(q), torch.zeros(B, H, 64, Tp, dtype=torch.bfloat16, device=DEV)
    ops.gemm_qkv(x.to(DEV).bfloat16(), W.to(DEV).bfloat16(), bias.to(DEV), B, T, q, k, vt, row_scale=rs.to(DEV))
    ref = ((x.double() @ W.double().T) * rs[:, None].double() + bias.double()).reshape(B, T, 3, H, 64)
    qs = 0.125 * 1.4426950408889634
    np.testing.assert_allclose(q[:, :, :T].float().cpu().numpy(), (ref[:, :, 0].permute(0, 2, 1, 3) * qs).numpy(), rtol=8e-3, atol=8e-3)
    np.testing.assert_allclose(k[:, :, :T].float().cpu().numpy(), ref[:, :, 1].permute(0, 2, 1, 3).numpy(), rtol=8e-3, atol=8e-3)
    np.testing.assert_allclose(vt[:, :, :, :T].float().cpu().numpy(), ref[:, :, 2].permute(0, 2, 3, 1).numpy(), rtol=8e-3, atol=8e-3)
    assert not q[:, :, T:].any() and not k[:, :, T:].any() and not vt[:, :, :, T:].any()     # padding is never written


def test_ln_fold_weights_reproduce_layernorm_then_linear():
    """runtime.ln_fold in float64: rstd * (x . W''^T) + b' == LN(x) . W^T + b, whatever the row mean."""
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(50, 384, generator=g) * 2 + 5).double()
    W, b = torch.randn(96, 384, generator=g), torch.randn(96, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(384, generator=g), 0.2 * torch.randn(384, generator=g)
    Wf, bf = runtime.ln_fold(W, b, gamma, beta)
    ref = torch.nn.functional.layer_norm(x, (384,), gamma.double(), beta.double(), 1e-5) @ W.double().T + b.double()
    rstd = 1.0 / torch.sqrt(x.var(dim=1, unbiased=False) + 1e-5)
    got = rstd[:, None] * (x @ Wf.double().T) + bf.double()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def _chain(kind, fold, unfused, monkeypatch, n_blocks=4):
    """n consecutive blocks of a network on a random residual stream, flags chained as the Packed* classes do."""
    from cmdiad_amd.runtime import _QkvBuffers, _pack_block, block_flags, transformer_block, transformer_block_unfused
    monkeypatch.setenv("CMDIAD_LN_FOLD", "1" if fold else "0")
    if kind == "vit":
        seed, fmt, B, T, C, H, eps, qkv_bias, with_pos, read_after = 31, "blocks.{}.", 2, 785, 768, 12, 1e-6, True, False, ()
    else:
        seed, fmt, B, T, C, H, eps, qkv_bias, with_pos, read_after = 21, "blocks.blocks.{}.", 3, 1024, 384, 6, 1e-5, False, True, (1,)
    sd = nets.synth_state_dict(kind, seed)
    blocks = [_pack_block(sd, fmt.format(i), DEV, qkv_bias) for i in range(n_blocks)]
    assert ("qkv_wf" in blocks[0]) == fold
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B * T, C, generator=g).to(DEV)
    pos = 0.1 * torch.randn(B * T, C, generator=g).to(DEV) if with_pos else None
    bufs, state, taps = _QkvBuffers(), {}, []
    for i, blk in enumerate(blocks):
        fl = block_flags(i, n_blocks, fold, read_after)
        if unfused:
            transformer_block_unfused(x, blk, B, T, H, eps, bufs, pos=pos, flags=fl, state=state)
        else:
            transformer_block(x, blk, B, T, H, eps, bufs, pos=pos, flags=fl)
        if i in read_after:
            taps.append(x.clone())
    return x, taps, (sd, fmt, B, T, C, H, eps, pos)


@pytest.mark.parametrize("kind", ["vit", "pointmae"])
def test_block_entry_point_equals_its_launches_with_the_fold_chained(kind, monkeypatch):
    for fold in (True, False):
        xa, ta, _ = _chain(kind, fold, False, monkeypatch)
        xb, tb, _ = _chain(kind, fold, True, monkeypatch)
        assert torch.equal(xa, xb) and torch.isfinite(xa).all()
        assert all(torch.equal(a, b) for a, b in zip(ta, tb))


@pytest.mark.parametrize("kind", ["vit", "pointmae"])
def test_folded_blocks_vs_separate_layernorm_and_fp32(kind, monkeypatch):
    """Four chained blocks: the folded form against the separate-LayerNorm form and both against torch in fp32 -- the fold must
    not cost accuracy (its error against fp32 within 15 % of the unfused form's) and the two agree to the bf16 noise floor."""
    xf, tf, (sd, fmt, B, T, C, H, eps, pos) = _chain(kind, True, False, monkeypatch)
    xu, tu, _ = _chain(kind, False, False, monkeypatch)
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B * T, C, generator=g).reshape(B, T, C)
    p = pos.cpu().reshape(B, T, C) if pos is not None else None
    with torch.no_grad():
        for i in range(4):
            x = nets._block(x + p if p is not None else x, sd, fmt.format(i)[:-1], H, eps)
    ref = x.reshape(B * T, C)
    scale = ref.abs().mean().item()
    ef, eu = (xf.cpu() - ref).abs().mean().item() / scale, (xu.cpu() - ref).abs().mean().item() / scale
    d = (xf - xu).abs().mean().item() / scale
    assert ef < 0.01 and eu < 0.01 and ef < 1.15 * eu + 1e-4 and d < 0.01, (ef, eu, d)
    for a, b in zip(tf, tu):       # the fetch-layer outputs too (read before the next block's pos is added)
        assert (a - b).abs().mean().item() / scale < 0.01


def test_fold_flags_are_validated():
    from cmdiad_amd._native import NativeError
    from cmdiad_amd.runtime import _QkvBuffers, _pack_block
    import os
    os.environ["CMDIAD_LN_FOLD"] = "0"
    try:
        blk = _pack_block(nets.synth_state_dict("pointmae", 21), "blocks.blocks.0.", DEV, False)
    finally:
        del os.environ["CMDIAD_LN_FOLD"]
    x = torch.randn(1024, 384, device=DEV)
    with pytest.raises(NativeError, match="folded"):
        runtime.transformer_block(x, blk, 1, 1024, 6, 1e-5, _QkvBuffers(), flags=ops.BLOCK_PREP_NEXT)
    A = torch.randn(128, 64, device=DEV).bfloat16()
    with pytest.raises(NativeError, match="ln_xb"):
        ops.gemm(A, A[:64].contiguous(), bias=torch.zeros(64, device=DEV), want_f32=True, want_bf16=False,
                 ln_xb=torch.empty(128, 64, dtype=torch.bfloat16, device=DEV), ln_part=torch.empty(1, 128, 2, device=DEV))
    with pytest.raises(ValueError, match="row_scale"):
        ops.gemm(A, A[:64].contiguous(), row_scale=torch.ones(128, device=DEV))
