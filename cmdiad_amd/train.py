"""Training path of the FtoF distillation network on the HIP kernels.

``direction_loss`` is what ``HallucinationCrossModalityNetwork.forward`` calls: one direction
(LayerNorm -> fc1 -> GELU -> fc2 -> GELU -> fc3 -> GELU -> loss against the real features of the other
modality) as a ``torch.autograd.Function`` over the eight parameters of that direction, so the reference
trainer's ``loss.backward()`` + ``torch.optim.Adam`` (hallucination_network_pretrain.py:148-154, :261) work
unchanged.  Forward and backward are computed together by the kernels:

  forward   h0 = LN(x) [bf16] ; z1 = h0 W1^T + b1 ; a1 = GELU(z1) ; z2 = a1 W2^T + b2 ; a2 = GELU(z2) ;
            z3 = a2 W3^T + b3 (fp32) ; y = GELU(z3) ; loss                       (cmdiad_gemm_bf16, cmdiad_loss_head)
  (mlp_depth > 1, utils/utils.py:103-115: that fc1..GELU chain repeated, each block's closing GELU feeding the next)
  backward  dz3 = dL/dy * GELU'(z3) (fused in the loss head) ; da2 = dz3 W3 (* GELU'(z2) fused in the
            GEMM epilogue) ; da1 likewise ; dh0 = dz1 W1 ; dW_k = dz_k^T a_{k-1} as split-K GEMMs over
            the token dimension with fixed-order slab reduction ; db_k = column sums ; LayerNorm
            gamma/beta gradients from dh0 and the saved row statistics.

``FusedAdam`` is an optional torch.optim.Optimizer with the update of torch.optim.Adam (no weight decay,
no amsgrad) running in one HIP kernel per parameter.
"""
import os

import torch

from . import _native as nat
from . import ops

_MODE = {"l2": 0, "cos_dist": 1, "smooth_l1": 2}
SPLIT_K = 8
CHUNKS = 64


def _call(name, *args):
    nat.check(getattr(nat.lib(), name)(*args), name)


def _reduce_slabs(slabs, S, n, out, scale=1.0):
    _call("cmdiad_reduce_slabs", ops._p(slabs), S, n, n, float(scale), ops._p(out), ops._stream())
    return out


def _dw(dz, a, out_shape):
    """(dW [dout, din], db [dout]) = sums over the M tokens of dz[m, :]^T a[m, :] and of dz[m, :], split-K over M.  Both
    operands stay row-major (cmdiad_gemm_tn_bf16 gathers its MFMA fragments with transposing LDS reads and adds the column
    sums of dz from the tiles it has staged anyway); CMDIAD_TRAIN_TN=0 selects the earlier form -- transpose both operands,
    the K-contiguous GEMM, a separate column-sum kernel -- for A/B runs."""
    dzp, ap = _pad_rows(dz), _pad_rows(a)
    M = dzp.shape[0]
    split = SPLIT_K if M >= 64 * SPLIT_K * 4 else 1
    if os.environ.get("CMDIAD_TRAIN_TN", "1") != "0":
        slabs, cs = ops.gemm_tn(dzp, ap, split_k=split, want_colsum=True)
        db = cs if split == 1 else _reduce_slabs(cs, split, cs.shape[1], torch.empty((cs.shape[1],), dtype=torch.float32, device=dz.device))
    else:
        slabs, _ = ops.gemm(ops.transpose_bf16(dzp), ops.transpose_bf16(ap), want_f32=True, want_bf16=False, split_k=split)
        db = _db(dz)
    if split == 1:
        return slabs, db
    out = torch.empty(out_shape, dtype=torch.float32, device=dz.device)
    return _reduce_slabs(slabs, split, out.numel(), out), db


def _db(dz):
    M, N = dz.shape
    part = torch.empty((CHUNKS, N), dtype=torch.float32, device=dz.device)
    _call("cmdiad_colsum_bf16", ops._p(dz), M, N, CHUNKS, ops._p(part), ops._stream())
    out = torch.empty((N,), dtype=torch.float32, device=dz.device)
    return _reduce_slabs(part, CHUNKS, N, out)


def _pad_rows(t, mult=64):
    """K of the dW GEMMs is the token count: pad to a multiple of 64 with zero rows (they add nothing)."""
    M = t.shape[0]
    if M % mult == 0:
        return t
    pad = torch.zeros((mult - M % mult, t.shape[1]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], 0)


def forward_backward(x, target, params, dist_method, batch, need_grad=True):
    """x, target [M, D] f32 cuda; params = (ln_w, ln_b) + (w1, b1, w2, b2, w3, b3) per MlpBlock of the direction (mlp_depth
    of them, utils/utils.py:103-115: the blocks are chained, each ends in a GELU) -- fp32 cuda tensors.
    Returns (loss 0-dim tensor, grads tuple in the order of params | None): loss = sum_rows(...) / batch."""
    ln_w, ln_b = params[:2]
    blocks = [params[2 + 6 * d: 8 + 6 * d] for d in range((len(params) - 2) // 6)]
    assert blocks and len(params) == 2 + 6 * len(blocks)
    M, D = x.shape
    dev = x.device
    mode = _MODE[dist_method]
    mean = torch.empty((M,), dtype=torch.float32, device=dev)
    rstd = torch.empty((M,), dtype=torch.float32, device=dev)
    x = x.contiguous()
    h0 = ops.layernorm(x, ln_w, ln_b, 1e-5, stats=(mean, rstd))
    saved = []          # per block: (input bf16, z1, a1, z2, a2, z3 bf16 | None, (w1h, w2h, w3h))
    h = h0
    z3 = None
    for d, (w1, b1, w2, b2, w3, b3) in enumerate(blocks):
        last = d == len(blocks) - 1
        w1h, w2h, w3h = ops.cast_bf16(w1.contiguous()), ops.cast_bf16(w2.contiguous()), ops.cast_bf16(w3.contiguous())
        H = w1.shape[0]
        z1 = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
        z2 = torch.empty((M, H), dtype=torch.bfloat16, device=dev)
        _, a1 = ops.gemm(h, w1h, bias=b1, act=ops.ACT_GELU, out_pre_bf16=z1)
        _, a2 = ops.gemm(a1, w2h, bias=b2, act=ops.ACT_GELU, out_pre_bf16=z2)
        if last:   # the block's closing GELU (and its derivative) is fused into the loss head, on the fp32 pre-activation
            z3, _ = ops.gemm(a2, w3h, bias=b3, want_f32=True, want_bf16=False)
            saved.append((h, z1, a1, z2, a2, None, (w1h, w2h, w3h)))
        else:
            z3b = torch.empty((M, w3.shape[0]), dtype=torch.bfloat16, device=dev)
            _, nxt = ops.gemm(a2, w3h, bias=b3, act=ops.ACT_GELU, out_pre_bf16=z3b)
            saved.append((h, z1, a1, z2, a2, z3b, (w1h, w2h, w3h)))
            h = nxt
    Dout = blocks[-1][4].shape[0]
    row_loss = torch.empty((M,), dtype=torch.float32, device=dev)
    dz3 = torch.empty((M, Dout), dtype=torch.bfloat16, device=dev) if need_grad else None
    _call("cmdiad_loss_head", ops._p(z3), ops._p(target.contiguous()), M, Dout, mode, 1.0 / batch, ops._p(row_loss),
          ops._p(dz3), None, ops._stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _call("cmdiad_sum_vector", ops._p(row_loss), M, 1.0 / batch, ops._p(loss), ops._stream())
    if not need_grad:
        return loss, None
    # ---- backward, last block first
    grads = [None] * len(blocks)
    dh0 = None
    for d in range(len(blocks) - 1, -1, -1):
        inp, z1, a1, z2, a2, _, (w1h, w2h, w3h) = saved[d]
        w1, _, w2, _, w3, _ = blocks[d]
        w3t, w2t, w1t = ops.transpose_bf16(w3h), ops.transpose_bf16(w2h), ops.transpose_bf16(w1h)   # [H,Dout], [H,H], [D,H]
        _, dz2 = ops.gemm(dz3, w3t, dact_of=z2)                       # [M,H]  = (dz3 W3) * GELU'(z2)
        _, dz1 = ops.gemm(dz2, w2t, dact_of=z1)                       # [M,H]
        g_w3, g_b3 = _dw(dz3, a2, w3.shape)
        g_w2, g_b2 = _dw(dz2, a1, w2.shape)
        g_w1, g_b1 = _dw(dz1, inp, w1.shape)
        grads[d] = (g_w1, g_b1, g_w2, g_b2, g_w3, g_b3)
        if d > 0:   # into the previous block through its closing GELU: (dz1 W1) * GELU'(z3 of block d-1)
            _, dz3 = ops.gemm(dz1, w1t, dact_of=saved[d - 1][5])
        else:
            dh0, _ = ops.gemm(dz1, w1t, want_f32=True, want_bf16=False)   # [M,D] f32
    pg = torch.empty((CHUNKS, D), dtype=torch.float32, device=dev)
    pb = torch.empty((CHUNKS, D), dtype=torch.float32, device=dev)
    _call("cmdiad_ln_param_grad", ops._p(dh0), ops._p(x), ops._p(mean), ops._p(rstd), M, D, CHUNKS, ops._p(pg), ops._p(pb),
          ops._stream())
    g_lnw = _reduce_slabs(pg, CHUNKS, D, torch.empty((D,), dtype=torch.float32, device=dev))
    g_lnb = _reduce_slabs(pb, CHUNKS, D, torch.empty((D,), dtype=torch.float32, device=dev))
    return loss, (g_lnw, g_lnb) + tuple(g for blk in grads for g in blk)


class _DirectionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, dist_method, batch, need_grad, *params):
        loss, grads = forward_backward(x, target, tuple(p.detach() for p in params), dist_method, batch, need_grad)
        ctx.grads, ctx.n_params = grads, len(params)
        return loss

    @staticmethod
    def backward(ctx, g):
        grads = ctx.grads
        if grads is None:
            return (None,) * (5 + ctx.n_params)
        return (None, None, None, None, None) + tuple(gr * g for gr in grads)


def direction_params(module, src):
    norm = getattr(module, f"{src}_norm")
    out = [norm.weight, norm.bias]
    for mlp in getattr(module, f"{src}_mlp").mlp_module:       # mlp_depth chained MlpBlocks (utils/utils.py:103-115)
        out += [mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, mlp.fc3.weight, mlp.fc3.bias]
    return tuple(out)


def direction_loss(module, src, x, target, dist_method="l2"):
    """src in {'xyz','rgb'}: features of modality `src` [B,T,D] -> loss against `target` [B,T,D'] (sum / B)."""
    if dist_method not in _MODE:
        raise NotImplementedError(dist_method)
    params = direction_params(module, src)
    dev = params[0].device
    B = x.shape[0]
    x2 = x.to(dev).float().reshape(-1, x.shape[-1])
    t2 = target.to(dev).float().reshape(-1, target.shape[-1])
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)  # (always False inside Function.forward)
    return _DirectionLoss.apply(x2, t2, dist_method, B, need, *params)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-8) semantics, one HIP kernel per parameter."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p)
                    st["v"] = torch.zeros_like(p)
                st["step"] += 1
                _call("cmdiad_adam_step", ops._p(p), ops._p(p.grad.contiguous()), ops._p(st["m"]), ops._p(st["v"]), p.numel(),
                      float(group["lr"]), b1, b2, group["eps"], st["step"], 1.0, None, ops._stream())
                p.view(-1)[:0].zero_()  # the kernel wrote p in place: bump its version so packed caches refresh
