"""Training path of the convolutional FtoF head on the HIP kernels (reference
models/hallucination_network.py:72-147, trained by hallucination_network_pretrain.py:106-147 in train() mode).

``tower_loss`` is what ``HallucinationCrossModalityConv.forward`` calls per direction: conv3x3 -> BatchNorm2d (batch statistics)
-> ReLU three times, conv3x3, mean row norm against the other modality's real features -- as a ``torch.autograd.Function``
over the tower's ten parameters, so the reference trainer's ``loss.backward()`` + ``torch.optim.Adam`` work unchanged.

  forward   x0 = bf16(tokens) [B,56,56,C] (the token layout IS NHWC: feature_reshape, hallucination_network.py:10-14, only renames it)
            z_l = conv3x3(x_l, W_l) fp32 (cmdiad_conv2d_nhwc_bf16) ; (mean, var)_l = column moments of z_l (cmdiad_col_moments) ;
            x_{l+1} = relu(z_l * scale_l + shift_l) bf16 (cmdiad_bn_relu_fwd) ; y = z_3 ; loss head (cmdiad_loss_head, no output
            activation / sigmoid of both sides)
  backward  dz_3 = dL/dy (loss head) ; per layer, last first:
            dW_l[n,c,ky,kx] = sum_m dz_l[m,n] x_l[m shifted by the tap, c]: nine cmdiad_gemm_tn_bf16 products over zero-bordered
            copies, in which a tap's shift is a row offset ; dx_l = conv3x3(dz_l, W_l flipped and transposed) fp32 ;
            (dz_{l-1}, dgamma, dbeta) = BatchNorm + ReLU backward (cmdiad_bn_relu_bwd_*).
The BatchNorm running statistics are updated as torch does (momentum 0.1, unbiased variance, num_batches_tracked)."""
import torch

from . import _native as nat
from . import ops
from .train import SPLIT_K, _reduce_slabs

EPS = 1e-5   # nn.BatchNorm2d default (hallucination_network.py:81)


def _call(name, *args):
    nat.check(getattr(nat.lib(), name)(*args), name)


def _conv_w(w):
    """[N,C,3,3] f32 -> tap-major bf16 [N, 9 C] (the layout of cmdiad_conv2d_nhwc_bf16)."""
    return ops.cast_bf16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())


def _conv_w_dgrad(w):
    """Weights of the data-gradient convolution: dx[y,x,c] = sum_{ky',kx',n} dz[y+ky'-1, x+kx'-1, n] w[n,c,2-ky',2-kx']."""
    return ops.cast_bf16(w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous())


def _wgrad(dz, x, B, H, W):
    """dz [M,N] bf16, x [M,C] bf16 (M = B*H*W, NHWC rows) -> dW [N,C,3,3] f32."""
    N, C = dz.shape[1], x.shape[1]
    dzp, g, rows = ops.pad_nhwc(dz.view(B, H, W, N))
    xp, gx, _ = ops.pad_nhwc(x.view(B, H, W, C))
    P = dzp[g:g + rows]
    split = SPLIT_K if rows >= 64 * SPLIT_K * 4 else 1
    taps = torch.empty((9, N, C), dtype=torch.float32, device=dz.device)
    for ky in range(3):
        for kx in range(3):
            off = (ky - 1) * (W + 2) + (kx - 1)
            Q = xp[gx + off:gx + off + rows]
            out = ops.gemm_tn(P, Q, split_k=split)
            if split == 1:
                taps[ky * 3 + kx].copy_(out)
            else:
                _reduce_slabs(out, split, N * C, taps[ky * 3 + kx])
    return taps.permute(1, 2, 0).reshape(N, C, 3, 3).contiguous()


def forward_backward(x, target, params, sigmoid, batch, need_grad=True):
    """x [B,T,C] f32, target [B,T,N] f32 (T = H*W tokens of a square map), params = (w0, g0, b0, w1, g1, b1, w2, g2, b2, w3).
    -> (loss 0-dim, grads tuple in the order of params | None, [(batch mean, biased batch variance) per BatchNorm] in float64)."""
    B, T, C = x.shape
    H = W = int(round(T ** 0.5))
    assert H * W == T
    dev = x.device
    M = B * T
    ws = [params[0], params[3], params[6], params[9]]
    bns = [(params[1], params[2]), (params[4], params[5]), (params[7], params[8])]
    xs = [ops.cast_bf16(x.reshape(M, C).contiguous())]
    zs, stats, affine = [], [], []
    for l in range(3):
        N = ws[l].shape[0]
        z, _ = ops.conv2d_nhwc(xs[l].view(B, H, W, -1), _conv_w(ws[l]), N, want_f32=True, want_bf16=False)
        z = z.view(M, N)
        mean64, var64 = ops.col_moments(z)
        rstd = (1.0 / torch.sqrt(var64 + EPS)).float()
        mean = mean64.float()
        scale = (bns[l][0].double() * (1.0 / torch.sqrt(var64 + EPS))).float()
        shift = (bns[l][1].double() - mean64 * scale.double()).float()
        xs.append(ops.bn_relu_fwd(z, scale, shift))
        zs.append(z); stats.append((mean64, var64)); affine.append((scale, shift, mean, rstd))
    Nout = ws[3].shape[0]
    y, _ = ops.conv2d_nhwc(xs[3].view(B, H, W, -1), _conv_w(ws[3]), Nout, want_f32=True, want_bf16=False)
    y = y.view(M, Nout)
    row_loss = torch.empty((M,), dtype=torch.float32, device=dev)
    dz = torch.empty((M, Nout), dtype=torch.bfloat16, device=dev) if need_grad else None
    mode = 0 + (512 if sigmoid else 256)   # l2 rows, CMDIAD_LOSS_OUT_SIGMOID / CMDIAD_LOSS_OUT_NONE
    _call("cmdiad_loss_head", ops._p(y), ops._p(target.reshape(M, Nout).contiguous()), M, Nout, mode, 1.0 / batch, ops._p(row_loss),
          ops._p(dz), None, ops._stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _call("cmdiad_sum_vector", ops._p(row_loss), M, 1.0 / batch, ops._p(loss), ops._stream())
    if not need_grad:
        return loss, None, stats
    grads = [None] * 10
    for l in range(3, -1, -1):
        grads[3 * l] = _wgrad(dz, xs[l], B, H, W)
        if l == 0:
            break
        dx, _ = ops.conv2d_nhwc(dz.view(B, H, W, -1), _conv_w_dgrad(ws[l]), ws[l].shape[1], want_f32=True, want_bf16=False)
        scale, shift, mean, rstd = affine[l - 1]
        dz, dgamma, dbeta = ops.bn_relu_bwd(dx.view(M, -1), zs[l - 1], scale, shift, mean, rstd)
        grads[3 * (l - 1) + 1], grads[3 * (l - 1) + 2] = dgamma, dbeta
    return loss, tuple(grads), stats


class _TowerLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, sigmoid, batch, need_grad, tower, *params):
        loss, grads, stats = forward_backward(x, target, tuple(p.detach() for p in params), sigmoid, batch, need_grad)
        M = x.shape[0] * x.shape[1]
        bn_layers = [m for m in tower if isinstance(m, torch.nn.BatchNorm2d)]
        with torch.no_grad():   # what nn.BatchNorm2d does in train(): momentum 0.1, UNBIASED variance into the running buffer
            for bn, (mean64, var64) in zip(bn_layers, stats):
                if bn.track_running_stats and bn.running_mean is not None:
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean64.to(bn.running_mean.dtype), alpha=mom)
                    bn.running_var.mul_(1 - mom).add_((var64 * (M / (M - 1))).to(bn.running_var.dtype), alpha=mom)
                    bn.num_batches_tracked += 1
        ctx.grads, ctx.n_params = grads, len(params)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            return (None,) * (6 + ctx.n_params)
        return (None,) * 6 + tuple(gr * g for gr in ctx.grads)


def tower_params(tower):
    convs = [m for m in tower if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in tower if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(convs) == 4 and len(bns) == 3 and all(c.bias is None and c.kernel_size == (3, 3) for c in convs)
    out = []
    for l in range(3):
        out += [convs[l].weight, bns[l].weight, bns[l].bias]
    return tuple(out + [convs[3].weight])


def tower_loss(tower, x, target, sigmoid):
    """tower: the nn.Sequential of one direction; x [B,3136,C] features of its input modality, target [B,3136,768] the real
    features of the other -> sum over rows of ||tower(x) - target||_2 / B (sigmoid: of both sides first)."""
    params = tower_params(tower)
    dev = params[0].device
    x, target = x.to(dev).float().contiguous(), target.to(dev).float().contiguous()
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _TowerLoss.apply(x, target, bool(sigmoid), x.shape[0], need, tower, *params)
