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
import os

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


def _split_for(rows, n1, n2):
    """Slices of the token dimension for a weight-gradient product whose output is only a few 128 x 128 tiles: enough workgroups to
    fill the chip (~512: two co-resident workgroups per CU; 36 tiles x 8 slices = 288 left the 768 x 768 products of the conv head at
    505 TFLOP/s), at least two 64-row steps per slice."""
    tiles = ((n1 + 127) // 128) * ((n2 + 127) // 128)
    target = int(os.environ.get("CMDIAD_WGRAD_BLOCKS", "512"))     # two co-resident workgroups per CU
    return int(max(1, min(max(1, target // tiles), rows // 128, 64)))


def _wgrad(dz, x, B, H, W):
    """dz [M,N] bf16, x [M,C] bf16 (M = B*H*W, NHWC rows) -> dW [N,C,3,3] f32."""
    N, C = dz.shape[1], x.shape[1]
    dzp, g, rows = ops.pad_nhwc(dz.view(B, H, W, N))
    xp, gx, _ = ops.pad_nhwc(x.view(B, H, W, C))
    P = dzp[g:g + rows]
    split = _split_for(rows, N, C)
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
        (scale, shift, mean, rstd), (mean64, var64) = _bn_affine(z, bns[l][0].contiguous(), bns[l][1].contiguous())
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


def _check_bn(bns):
    """The hand-written BatchNorm forward / backward is the affine batch-statistics form with eps = EPS (nn.BatchNorm2d's default,
    what every BatchNorm of the reference's heads is constructed with): anything else must not be trained silently differently."""
    for bn in bns:
        if not bn.affine or abs(float(bn.eps) - EPS) > 1e-12:
            raise NotImplementedError(f"hand-written BatchNorm training path: affine BatchNorm2d with eps = {EPS} only (got eps = {bn.eps}, "
                                      f"affine = {bn.affine}); use the module's torch layers (CMDIAD_HRNET_TRAIN=torch / CMDIAD_CONV_TRAIN=torch)")


def tower_params(tower):
    convs = [m for m in tower if isinstance(m, torch.nn.Conv2d)]
    bns = [m for m in tower if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(convs) == 4 and len(bns) == 3 and all(c.bias is None and c.kernel_size == (3, 3) for c in convs)
    _check_bn(bns)
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


# ------------------------------------------------------------------------------------------------------------------------
# Feature-to-input convolutional head (reference models/hallucination_network.py:185-220): conv 768 -> 384 on the 56 x 56 map,
# bicubic x4, conv 384 -> 96 + ReLU, conv 96 -> 32 + ReLU, conv 32 -> 3; loss = sum over pixels of the 3-vector norm of
# (hallucination - image) / B (:211-220).  Channel counts that are not multiples of 64 are zero-padded (96 -> 128, 32 -> 64,
# 3 -> 4 for outputs / 64 where the tensor is a convolution operand), as in runtime.PackedFtoIConv: padded weights and biases are
# zero, so padded activations and their gradients stay zero.
# ------------------------------------------------------------------------------------------------------------------------
def _pad_w(w, n_pad, c_pad):
    N, C = w.shape[:2]
    out = torch.zeros((n_pad, c_pad, 3, 3), dtype=torch.float32, device=w.device)
    out[:N, :C] = w
    return out


def _pad_v(b, n_pad):
    out = torch.zeros((n_pad,), dtype=torch.float32, device=b.device)
    out[:b.shape[0]] = b
    return out


def _wgrad_bias(dz, x, B, H, W):
    """As _wgrad, plus the bias gradient (column sums of dz, from the tiles the centre tap's product stages anyway)."""
    N, C = dz.shape[1], x.shape[1]
    dzp, g, rows = ops.pad_nhwc(dz.view(B, H, W, N))
    xp, gx, _ = ops.pad_nhwc(x.view(B, H, W, C))
    P = dzp[g:g + rows]
    split = _split_for(rows, N, C)
    taps = torch.empty((9, N, C), dtype=torch.float32, device=dz.device)
    db = torch.empty((N,), dtype=torch.float32, device=dz.device)
    for t in range(9):
        off = (t // 3 - 1) * (W + 2) + (t % 3 - 1)
        Q = xp[gx + off:gx + off + rows]
        if t == 4:
            out, cs = ops.gemm_tn(P, Q, split_k=split, want_colsum=True)
            if split == 1:
                db.copy_(cs)
            else:
                _reduce_slabs(cs, split, N, db)
        else:
            out = ops.gemm_tn(P, Q, split_k=split)
        if split == 1:
            taps[t].copy_(out)
        else:
            _reduce_slabs(out, split, N * C, taps[t])
    return taps.permute(1, 2, 0).reshape(N, C, 3, 3).contiguous(), db


def ftoi_forward_backward(feature, img, params, batch, need_grad=True):
    """feature [B,3136,768] f32, img [B,3,224,224] f32, params = (w1, b1, w2, b2, w3, b3, w4, b4) of conv1..conv4.
    -> (loss 0-dim, grads in the order of params | None)."""
    w1, b1, w2, b2, w3, b3, w4, b4 = params
    B, T, C = feature.shape
    h = int(round(T ** 0.5))
    H = img.shape[-1]
    dev = feature.device
    n1, n2, n3, n4 = w1.shape[0], w2.shape[0], w3.shape[0], w4.shape[0]             # 384, 96, 32, 3
    p2, p3, p4 = (n2 + 63) // 64 * 64, (n3 + 63) // 64 * 64, (n4 + 3) // 4 * 4      # 128, 64, 4
    W1, W2 = w1, _pad_w(w2, p2, n1)
    W3, W4 = _pad_w(w3, p3, p2), _pad_w(w4, p4, p3)
    x0 = ops.cast_bf16(feature.reshape(B * T, C).contiguous()).view(B, h, h, C)
    h1, _ = ops.conv2d_nhwc(x0, _conv_w(W1), n1, bias=b1.contiguous(), want_f32=True, want_bf16=False)               # [B,56,56,384] f32
    u = ops.upsample_bicubic(h1, n1, H, H)                                                                           # [B,224,224,384] bf16
    _, h2 = ops.conv2d_nhwc(u, _conv_w(W2), p2, bias=_pad_v(b2, p2), act=ops.ACT_RELU)                               # [.,128] bf16
    _, h3 = ops.conv2d_nhwc(h2, _conv_w(W3), p3, bias=_pad_v(b3, p3), act=ops.ACT_RELU)                              # [.,64]
    out, _ = ops.conv2d_nhwc(h3, _conv_w(W4), p4, bias=_pad_v(b4, p4), want_f32=True, want_bf16=False)               # [.,4] f32
    M = B * H * H
    target = torch.zeros((B, H, H, p4), dtype=torch.float32, device=dev)
    target[..., :n4] = img.permute(0, 2, 3, 1)
    row_loss = torch.empty((M,), dtype=torch.float32, device=dev)
    dout = torch.empty((M, p4), dtype=torch.bfloat16, device=dev) if need_grad else None
    _call("cmdiad_loss_head", ops._p(out), ops._p(target), M, p4, 0 + 256, 1.0 / batch, ops._p(row_loss), ops._p(dout), None, ops._stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _call("cmdiad_sum_vector", ops._p(row_loss), M, 1.0 / batch, ops._p(loss), ops._stream())
    if not need_grad:
        return loss, None
    # conv4: the output gradient as a 64-channel operand (4 live), weight gradient, data gradient, ReLU of conv3
    d4 = torch.zeros((M, 64), dtype=torch.bfloat16, device=dev)
    d4[:, :p4] = dout
    gw4, gb4 = _wgrad_bias(d4, h3.view(M, p3), B, H, H)
    W4d = torch.zeros((64, p3, 3, 3), dtype=torch.float32, device=dev)
    W4d[:p4] = W4
    dh3, _ = ops.conv2d_nhwc(d4.view(B, H, H, 64), _conv_w_dgrad(W4d), p3, want_f32=True, want_bf16=False)
    dz3 = ops.relu_bwd(dh3.view(M, p3), h3.view(M, p3))
    gw3, gb3 = _wgrad_bias(dz3, h2.view(M, p2), B, H, H)
    dh2, _ = ops.conv2d_nhwc(dz3.view(B, H, H, p3), _conv_w_dgrad(W3), p2, want_f32=True, want_bf16=False)
    dz2 = ops.relu_bwd(dh2.view(M, p2), h2.view(M, p2))
    gw2, gb2 = _wgrad_bias(dz2, u.view(M, n1), B, H, H)
    du, _ = ops.conv2d_nhwc(dz2.view(B, H, H, p2), _conv_w_dgrad(W2), n1, want_f32=True, want_bf16=False)
    dh1 = ops.upsample_bicubic_bwd(du.view(B, H, H, n1), h, h)                                                       # [B,56,56,384] f32
    gw1, gb1 = _wgrad_bias(ops.cast_bf16(dh1.view(B * T, n1)), x0.view(B * T, C), B, h, h)
    return loss, (gw1, gb1, gw2[:n2].contiguous(), gb2[:n2].contiguous(), gw3[:n3, :n2].contiguous(), gb3[:n3].contiguous(),
                  gw4[:n4, :n3].contiguous(), gb4[:n4].contiguous())


class _FtoILoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature, img, batch, need_grad, *params):
        loss, grads = ftoi_forward_backward(feature, img, tuple(p.detach() for p in params), batch, need_grad)
        ctx.grads, ctx.n_params = grads, len(params)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            return (None,) * (4 + ctx.n_params)
        return (None,) * 4 + tuple(gr * g for gr in ctx.grads)


def ftoi_conv_loss(module, feature, img):
    """module: HallucinationFeatureToInputConv; feature [B,3136,768], img [B,3,224,224] -> sum over pixels of
    ||hallucination - img||_2 / B, differentiable w.r.t. conv1..conv4 (``norm`` is not part of the reference's forward)."""
    params = tuple(p for i in range(1, 5) for p in (getattr(module, f"conv{i}").weight, getattr(module, f"conv{i}").bias))
    dev = params[0].device
    feature, img = feature.to(dev).float().contiguous(), img.to(dev).float().contiguous()
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FtoILoss.apply(feature, img, feature.shape[0], need, *params)


# ------------------------------------------------------------------------------------------------------------------------
# Feature-to-input MLP head (reference models/hallucination_network.py:146-182): LayerNorm, Linear 768 -> 1152 -> 384 -> 96 ->
# out_dim with GELU between, the out_dim-channel 56 x 56 map upsampled bicubically to 224 x 224, loss as above.  The 96-wide layer
# is zero-padded to 128 and out_dim to 4 (64 where the gradient is a GEMM operand), as in runtime.PackedFtoIMLP.
# ------------------------------------------------------------------------------------------------------------------------
def ftoi_mlp_forward_backward(x, img, params, batch, need_grad=True):
    """x [B,3136,768] f32, img [B,out_dim,224,224] f32, params = (ln_w, ln_b, w1, b1, w2, b2, w3, b3, w4, b4).
    -> (loss 0-dim, grads in the order of params | None)."""
    from .train import CHUNKS, _dw
    ln_w, ln_b, w1, b1, w2, b2, w3, b3, w4, b4 = params
    B, T, D = x.shape
    h = int(round(T ** 0.5))
    H = img.shape[-1]
    dev = x.device
    M = B * T
    n3, n4 = w3.shape[0], w4.shape[0]                          # 96, out_dim
    p3, p4 = (n3 + 63) // 64 * 64, 4
    W3 = torch.zeros((p3, w3.shape[1]), dtype=torch.float32, device=dev); W3[:n3] = w3
    W4 = torch.zeros((64, p3), dtype=torch.float32, device=dev); W4[:n4, :n3] = w4          # 64 rows: its transpose is a GEMM operand
    w1h, w2h, w3h, w4h = ops.cast_bf16(w1.contiguous()), ops.cast_bf16(w2.contiguous()), ops.cast_bf16(W3), ops.cast_bf16(W4)
    x2 = x.reshape(M, D).contiguous()
    mean = torch.empty((M,), dtype=torch.float32, device=dev)
    rstd = torch.empty((M,), dtype=torch.float32, device=dev)
    h0 = ops.layernorm(x2.clone(), ln_w, ln_b, 1e-5, stats=(mean, rstd))
    z1 = torch.empty((M, w1.shape[0]), dtype=torch.bfloat16, device=dev)
    z2 = torch.empty((M, w2.shape[0]), dtype=torch.bfloat16, device=dev)
    z3 = torch.empty((M, p3), dtype=torch.bfloat16, device=dev)
    _, a1 = ops.gemm(h0, w1h, bias=b1, act=ops.ACT_GELU, out_pre_bf16=z1)
    _, a2 = ops.gemm(a1, w2h, bias=b2, act=ops.ACT_GELU, out_pre_bf16=z2)
    _, a3 = ops.gemm(a2, w3h, bias=_pad_v(b3, p3), act=ops.ACT_GELU, out_pre_bf16=z3)
    out, _ = ops.gemm(a3, w4h[:p4].contiguous(), bias=_pad_v(b4, p4), want_f32=True, want_bf16=False)              # [M,4] f32
    up = ops.upsample_bicubic(out.view(B, h, h, p4), p4, H, H, nchw=True)                                           # [B,4,H,H] f32
    Mu = B * H * H
    pred = up.permute(0, 2, 3, 1).contiguous()                                                                      # NHWC rows
    target = torch.zeros((B, H, H, p4), dtype=torch.float32, device=dev)
    target[..., :n4] = img.permute(0, 2, 3, 1)
    row_loss = torch.empty((Mu,), dtype=torch.float32, device=dev)
    dup = torch.empty((Mu, p4), dtype=torch.bfloat16, device=dev) if need_grad else None
    _call("cmdiad_loss_head", ops._p(pred), ops._p(target), Mu, p4, 0 + 256, 1.0 / batch, ops._p(row_loss), ops._p(dup), None, ops._stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _call("cmdiad_sum_vector", ops._p(row_loss), Mu, 1.0 / batch, ops._p(loss), ops._stream())
    if not need_grad:
        return loss, None
    dout = ops.upsample_bicubic_bwd(dup.float().view(B, H, H, p4), h, h)                                            # [B,56,56,4] f32
    dz4 = torch.zeros((M, 64), dtype=torch.bfloat16, device=dev)
    dz4[:, :p4] = dout.view(M, p4).to(torch.bfloat16)
    g_w4, g_b4 = _dw(dz4, a3, (64, p3))
    _, dz3 = ops.gemm(dz4, ops.transpose_bf16(w4h), dact_of=z3)          # [M,128] = (dz4 W4) * GELU'(z3)
    g_w3, g_b3 = _dw(dz3, a2, (p3, w3.shape[1]))
    _, dz2 = ops.gemm(dz3, ops.transpose_bf16(w3h), dact_of=z2)
    g_w2, g_b2 = _dw(dz2, a1, tuple(w2.shape))
    _, dz1 = ops.gemm(dz2, ops.transpose_bf16(w2h), dact_of=z1)
    g_w1, g_b1 = _dw(dz1, h0, tuple(w1.shape))
    dh0, _ = ops.gemm(dz1, ops.transpose_bf16(w1h), want_f32=True, want_bf16=False)
    pg = torch.empty((CHUNKS, D), dtype=torch.float32, device=dev)
    pb = torch.empty((CHUNKS, D), dtype=torch.float32, device=dev)
    _call("cmdiad_ln_param_grad", ops._p(dh0), ops._p(x2), ops._p(mean), ops._p(rstd), M, D, CHUNKS, ops._p(pg), ops._p(pb), ops._stream())
    g_lnw = _reduce_slabs(pg, CHUNKS, D, torch.empty((D,), dtype=torch.float32, device=dev))
    g_lnb = _reduce_slabs(pb, CHUNKS, D, torch.empty((D,), dtype=torch.float32, device=dev))
    return loss, (g_lnw, g_lnb, g_w1, g_b1, g_w2, g_b2, g_w3[:n3].contiguous(), g_b3[:n3].contiguous(),
                  g_w4[:n4, :n3].contiguous(), g_b4[:n4].contiguous())


class _FtoIMlpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, img, batch, need_grad, *params):
        loss, grads = ftoi_mlp_forward_backward(x, img, tuple(p.detach() for p in params), batch, need_grad)
        ctx.grads, ctx.n_params = grads, len(params)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            return (None,) * (4 + ctx.n_params)
        return (None,) * 4 + tuple(gr * g for gr in ctx.grads)


def ftoi_mlp_loss(module, x, img):
    """module: HallucinationRGBFeatureToXYZInputMLP; x [B,3136,768], img [B,out_dim,224,224]."""
    lin = [m for m in module.mlp if isinstance(m, torch.nn.Linear)]
    params = (module.rgb_norm.weight, module.rgb_norm.bias) + tuple(p for m in lin for p in (m.weight, m.bias))
    dev = params[0].device
    x, img = x.to(dev).float().contiguous(), img.to(dev).float().contiguous()
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FtoIMlpLoss.apply(x, img, x.shape[0], need, *params)


# ------------------------------------------------------------------------------------------------------------------------
# Input-to-feature head (reference models/hrnet.py:146-177, 251-299: what HRNet.forward runs is a ResNet-style trunk): stem
# conv3x3/2 3 -> 64, conv3x3/2 64 -> 128 (each + BatchNorm + ReLU), twelve Bottlenecks (1x1 -> 128, 3x3, 1x1 -> 512, each + BatchNorm;
# ReLU after the first two and after the residual sum; the first block widens its identity through 1x1 + BatchNorm), 1x1
# final_layer 512 -> 768 with bias; loss = sum over tokens of ||tokens - feature||_2 / B.  All BatchNorms on batch statistics.
#   * 1x1 convolutions are GEMMs on the [M, C] token matrix (forward, data gradient on the transposed weight with the identity
#     path's gradient as the GEMM's residual, weight gradient on gemm_tn);
#   * the first stem convolution (3 input channels) is a GEMM on the 27-column im2col of the image (built by torch.unfold: data
#     movement), padded to 64 columns;
#   * a stride-2 convolution's gradients are the stride-1 ones of its output gradient with zeros stuffed between the pixels.
# ------------------------------------------------------------------------------------------------------------------------
def _bn_affine(z, gamma, beta, eps=EPS):
    """Batch statistics of z [M,C] -> ((scale, shift, mean, rstd) f32, (mean, biased variance) f64): cmdiad_col_moments +
    cmdiad_bn_affine (two launches and one memset per BatchNorm)."""
    M, C = z.shape
    acc = torch.zeros((2, C), dtype=torch.float64, device=z.device)
    _call("cmdiad_col_moments", ops._p(z), M, C, z.stride(0), ops._p(acc[0]), ops._p(acc[1]), ops._stream())
    st64 = torch.empty((2, C), dtype=torch.float64, device=z.device)
    aff = torch.empty((4, C), dtype=torch.float32, device=z.device)
    _call("cmdiad_bn_affine", ops._p(acc[0]), ops._p(acc[1]), ops._p(gamma), ops._p(beta), M, float(eps), C, ops._p(st64[0]), ops._p(st64[1]),
          ops._p(aff[0]), ops._p(aff[1]), ops._p(aff[2]), ops._p(aff[3]), ops._stream())
    return (aff[0], aff[1], aff[2], aff[3]), (st64[0], st64[1])


def _w1x1(w):
    return ops.cast_bf16(w.reshape(w.shape[0], w.shape[1]).contiguous())


def hrnet_forward_backward(img, feature, P, batch, need_grad=True):
    """img [B,3,224,224] f32, feature [B,3136,768] f32, P: dict name -> fp32 parameter tensor (the state_dict names of the trunk).
    -> (loss, {name: grad} | None, {bn name: (batch mean, biased batch variance)} in float64)."""
    from .train import _dw
    B = img.shape[0]
    dev = img.device
    stats, saved = {}, {}
    wcache = {}

    def w16(name):          # bf16 [N,C] of a 1x1 convolution, cast once per step
        if name not in wcache:
            wcache[name] = _w1x1(P[name])
        return wcache[name]

    def dw1x1(dz, a, shape):   # weight gradient of a 1x1 convolution without bias: dz^T a over the tokens (split over M, fixed-order sum)
        split = _split_for(dz.shape[0], shape[0], shape[1])
        if dz.shape[0] % 64:
            return _dw(dz, a, (shape[0], shape[1]))[0].view(shape)
        out = ops.gemm_tn(dz, a, split_k=split)
        if split > 1:
            out = _reduce_slabs(out, split, shape[0] * shape[1], torch.empty((shape[0], shape[1]), dtype=torch.float32, device=dz.device))
        return out.view(shape)

    def bn(name, z, residual=None, relu=True, want_f32=False):
        aff, st = _bn_affine(z, P[name + ".weight"], P[name + ".bias"])
        stats[name] = st
        saved[name] = (z, aff)
        return ops.bn_relu_fwd(z, aff[0], aff[1], residual=residual, relu=relu, want_bf16=True, want_f32=want_f32)

    # stem
    H1 = (img.shape[-1] + 1) // 2
    cols64 = ops.im2col3x3(img.contiguous(), stride=2, ld=64)       # [B*112*112, 64] bf16: 27 columns (c, ky, kx) + zero padding
    W1 = torch.zeros((64, 64), dtype=torch.float32, device=dev)
    W1[:, :27] = P["conv1.weight"].reshape(64, 27)
    z1, _ = ops.gemm(cols64, ops.cast_bf16(W1), want_f32=True, want_bf16=False)
    y1 = bn("bn1", z1)                                                                          # [B*112*112, 64] bf16
    z2, _ = ops.conv2d_nhwc(y1.view(B, H1, H1, 64), _conv_w(P["conv2.weight"]), 128, 3, 2, want_f32=True, want_bf16=False)
    H2 = z2.shape[1]
    M = B * H2 * H2
    x16 = bn("bn2", z2.view(M, 128))
    x32 = None
    blocks = [f"layer{l}.{i}" for l in (1, 2, 3) for i in range(4)]
    for b in blocks:
        down = (b + ".downsample.0.weight") in P
        xin = x16
        zc1, _ = ops.gemm(xin, w16(b + ".conv1.weight"), want_f32=True, want_bf16=False)
        t1 = bn(b + ".bn1", zc1)
        zc2, _ = ops.conv2d_nhwc(t1.view(B, H2, H2, -1), _conv_w(P[b + ".conv2.weight"]), P[b + ".conv2.weight"].shape[0], 3, 1,
                                 want_f32=True, want_bf16=False)
        t2 = bn(b + ".bn2", zc2.view(M, -1))
        zc3, _ = ops.gemm(t2, w16(b + ".conv3.weight"), want_f32=True, want_bf16=False)
        if down:
            zd, _ = ops.gemm(xin, w16(b + ".downsample.0.weight"), want_f32=True, want_bf16=False)
            _, identity = bn(b + ".downsample.1", zd, relu=False, want_f32=True)
        else:
            identity = x32
        assert identity is not None, f"{b}: a stage's first bottleneck must widen its identity (downsample), hrnet.py:167-171"
        x16, x32 = bn(b + ".bn3", zc3, residual=identity, relu=True, want_f32=True)
        saved[b] = (xin, t1, t2, x16)
    Wf = _w1x1(P["final_layer.weight"])
    out, _ = ops.gemm(x16, Wf, bias=P["final_layer.bias"].contiguous(), want_f32=True, want_bf16=False)
    N = out.shape[1]
    row_loss = torch.empty((M,), dtype=torch.float32, device=dev)
    dout = torch.empty((M, N), dtype=torch.bfloat16, device=dev) if need_grad else None
    _call("cmdiad_loss_head", ops._p(out), ops._p(feature.reshape(M, N).contiguous()), M, N, 0 + 256, 1.0 / batch, ops._p(row_loss),
          ops._p(dout), None, ops._stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _call("cmdiad_sum_vector", ops._p(row_loss), M, 1.0 / batch, ops._p(loss), ops._stream())
    if not need_grad:
        return loss, None, stats
    G = {}

    def bn_bwd(name, dy, masked=True):
        z, (scale, shift, mean, rstd) = saved[name]
        dz, dg, db = ops.bn_relu_bwd(dy, z, scale, shift, mean, rstd, masked=masked)
        G[name + ".weight"], G[name + ".bias"] = dg, db
        return dz

    G["final_layer.weight"], G["final_layer.bias"] = _dw(dout, x16, (N, Wf.shape[1]))
    G["final_layer.weight"] = G["final_layer.weight"].view(N, -1, 1, 1)
    dX, _ = ops.gemm(dout, ops.transpose_bf16(Wf), want_f32=True, want_bf16=False)             # [M,512] f32
    for b in reversed(blocks):
        xin, t1, t2, xout = saved[b]
        w1, w2, w3 = P[b + ".conv1.weight"], P[b + ".conv2.weight"], P[b + ".conv3.weight"]
        g32 = ops.relu_bwd(dX, xout, want_bf16=False, want_f32=True)[1]                        # through the ReLU after the residual sum
        dz3 = bn_bwd(b + ".bn3", g32, masked=False)
        G[b + ".conv3.weight"] = dw1x1(dz3, t2, w3.shape)
        dt2, _ = ops.gemm(dz3, ops.transpose_bf16(w16(b + ".conv3.weight")), want_f32=True, want_bf16=False)
        dz2 = bn_bwd(b + ".bn2", dt2)
        G[b + ".conv2.weight"] = _wgrad(dz2, t1, B, H2, H2)
        dt1, _ = ops.conv2d_nhwc(dz2.view(B, H2, H2, -1), _conv_w_dgrad(w2), w2.shape[1], want_f32=True, want_bf16=False)
        dz1 = bn_bwd(b + ".bn1", dt1.view(M, -1))
        G[b + ".conv1.weight"] = dw1x1(dz1, xin, w1.shape)
        if (b + ".downsample.0.weight") in P:
            wd = P[b + ".downsample.0.weight"]
            dzd = bn_bwd(b + ".downsample.1", g32, masked=False)
            G[b + ".downsample.0.weight"] = dw1x1(dzd, xin, wd.shape)
            dxa, _ = ops.gemm(dzd, ops.transpose_bf16(w16(b + ".downsample.0.weight")), want_f32=True, want_bf16=False)
            dX, _ = ops.gemm(dz1, ops.transpose_bf16(w16(b + ".conv1.weight")), residual=dxa, want_f32=True, want_bf16=False)
        else:
            dX, _ = ops.gemm(dz1, ops.transpose_bf16(w16(b + ".conv1.weight")), residual=g32, want_f32=True, want_bf16=False)
    # stem: conv2 has stride 2 -- its gradients are those of a stride-1 convolution whose output gradient has zeros between the pixels
    dz2s = bn_bwd("bn2", dX)
    up = torch.zeros((B, H1, H1, 128), dtype=torch.bfloat16, device=dev)
    up[:, ::2, ::2] = dz2s.view(B, H2, H2, 128)
    G["conv2.weight"] = _wgrad(up.view(B * H1 * H1, 128), y1, B, H1, H1)
    dy1, _ = ops.conv2d_nhwc(up, _conv_w_dgrad(P["conv2.weight"]), 64, want_f32=True, want_bf16=False)
    dz1s = bn_bwd("bn1", dy1.view(B * H1 * H1, 64))
    G["conv1.weight"] = dw1x1(dz1s, cols64, (64, 64))[:, :27].reshape(64, 3, 3, 3).contiguous()
    return loss, G, stats


# The trunk's step is ~500 launches; with its reductions parallelised (round 4) their device time at batch 8 is 9.7 ms, below the
# ~13 ms the host needs to issue them one by one.  The whole forward + backward is therefore captured ONCE per (shapes, parameter
# storages) as a HIP graph -- on the third step with that key, after two eager ones -- and replayed: inputs are copied into the
# graph's static buffers, the loss / gradients / batch statistics are the graph's static outputs (consumed before the next replay:
# autograd scales the gradients into new tensors, the running statistics are updated at once).  CMDIAD_HRNET_GRAPH=0: always eager.
_HRNET_GRAPHS = {}


def _hrnet_step(img, feature, P, batch, need_grad):
    if os.environ.get("CMDIAD_HRNET_GRAPH", "1") == "0" or not need_grad:
        return hrnet_forward_backward(img, feature, P, batch, need_grad) + (None,)
    key = (tuple(img.shape), tuple(feature.shape), img.device.index) + tuple(p.data_ptr() for p in P.values())
    ent = _HRNET_GRAPHS.get(key)
    if ent is None:
        if len(_HRNET_GRAPHS) >= 4:      # parameters were re-allocated (a new module / optimizer state load): drop the old captures
            _HRNET_GRAPHS.clear()
        ent = _HRNET_GRAPHS[key] = {"seen": 0}
    if "graph" not in ent:
        ent["seen"] += 1
        if ent["seen"] < 3:
            return hrnet_forward_backward(img, feature, P, batch, need_grad) + (None,)
        ent["img"], ent["feature"] = img.clone(), feature.clone()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                ent["out"] = hrnet_forward_backward(ent["img"], ent["feature"], P, batch, need_grad)
            ent["graph"] = graph
        except Exception as e:      # noqa: BLE001 -- capture is an optimisation: this key stays eager, the step itself must not fail
            import warnings
            warnings.warn(f"HRNet trunk step: HIP-graph capture failed ({type(e).__name__}: {e}); this shape runs eagerly from now on")
            ent["graph"] = None
            ent.pop("out", None), ent.pop("img", None), ent.pop("feature", None)
            torch.cuda.synchronize()
    if ent["graph"] is None:
        return hrnet_forward_backward(img, feature, P, batch, need_grad) + (None,)
    ent["img"].copy_(img)
    ent["feature"].copy_(feature)
    ent["graph"].replay()
    ent["gen"] = ent.get("gen", 0) + 1
    loss, G, stats = ent["out"]
    return loss.clone(), G, stats, (ent, ent["gen"])     # (the loss outlives the step in most training loops; the gradients do not)


class _HRNetLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, feature, batch, need_grad, module, names, *params):
        P = {n: p.detach() for n, p in zip(names, params)}
        loss, G, stats, ctx.replay = _hrnet_step(img, feature, P, batch, need_grad)
        M1, M2 = img.shape[0] * ((img.shape[-1] + 1) // 2) ** 2, feature.shape[0] * feature.shape[1]
        mods = dict(module.named_modules())
        with torch.no_grad():   # nn.BatchNorm2d in train(): momentum, UNBIASED variance into the running buffer
            for name, (mean64, var64) in stats.items():
                bn = mods[name]
                if bn.track_running_stats and bn.running_mean is not None:
                    n = M1 if name == "bn1" else M2
                    mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
                    bn.running_mean.mul_(1 - mom).add_(mean64.to(bn.running_mean.dtype), alpha=mom)
                    bn.running_var.mul_(1 - mom).add_((var64 * (n / (n - 1))).to(bn.running_var.dtype), alpha=mom)
                    bn.num_batches_tracked += 1
        ctx.grads = None if G is None else tuple(G.get(n) for n in names)
        ctx.n_params = len(params)
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grads is None:
            return (None,) * (6 + ctx.n_params)
        if ctx.replay is not None and ctx.replay[0].get("gen") != ctx.replay[1]:
            raise RuntimeError("HRNet training step: backward() of a forward whose gradients have been overwritten -- the hand-written "
                               "step keeps them in its HIP graph's buffers until the NEXT forward of the same shapes; call backward() "
                               "first, or set CMDIAD_HRNET_GRAPH=0")
        return (None,) * 6 + tuple(None if gr is None else gr * g for gr in ctx.grads)


def hrnet_loss(module, img, feature):
    """module: models.hrnet.HRNet (c = 512); img [B,3,224,224], feature [B,3136,768] -> its training loss (hrnet.py:290-299),
    differentiable w.r.t. every parameter its forward uses (layer4 is constructed but never run: no gradient, as in the reference)."""
    names, params = zip(*[(n, p) for n, p in module.named_parameters() if not n.startswith("layer4.")])
    _check_bn([m for n, m in module.named_modules() if isinstance(m, torch.nn.BatchNorm2d) and not n.startswith("layer4.")])
    dev = params[0].device
    img, feature = img.to(dev).float().contiguous(), feature.to(dev).float().contiguous()
    need = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _HRNetLoss.apply(img, feature, img.shape[0], need, module, names, *params)
