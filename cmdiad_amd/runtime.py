"""Device-side forward passes of the three networks on the hot path, composed from the HIP kernels
in ops.py.  Weights are packed once (bf16 GEMM operands, fp32 biases / LayerNorm parameters,
eval-mode BatchNorm folded into the 1x1 convolutions) from a ``state_dict`` that uses the
reference's parameter names, so real checkpoints and synthetic weights load the same way.

Reference anchors: ViT-B/8 models/models.py:35-53 (timm VisionTransformer [external]);
Point-MAE models/models.py:183-243, 352-373; hallucination MLP models/hallucination_network.py:34-45,
utils/utils.py:86-115.
"""
import os

import torch

from . import ops


def _dev(t, device, dtype=torch.float32):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def _bf(t, device):
    return ops.cast_bf16(_dev(t, device)) if t.numel() % 4 == 0 else _dev(t, device).to(torch.bfloat16)


def ln_fold(weight, bias, gamma, beta):
    """LayerNorm folded into the Linear that follows it (csrc/gemm.hip "LayerNorm fold"):  LN(x) . W^T + b  =
    rstd * (x . W''^T) + b'  with  W''[n,k] = gamma[k] W[n,k] - mean_k(gamma[k] W[n,k])  (centred over k, so the row mean of x
    drops out of the product) and  b' = b + W beta.  -> (W'' fp32 [N,K], b' fp32 [N]); computed in float64."""
    W = weight.detach().double().cpu()
    Wg = W * gamma.detach().double().cpu()[None, :]
    Wc = Wg - Wg.mean(dim=1, keepdim=True)
    b0 = bias.detach().double().cpu() if bias is not None else torch.zeros(W.shape[0], dtype=torch.float64)
    return Wc.float(), (b0 + W @ beta.detach().double().cpu()).float()


def ln_fold_enabled(net=""):
    """Which networks run with their LayerNorms folded into the neighbouring products.  CMDIAD_LN_FOLD = pmae (default) | vit | 1
    (both) | 0 (none: every LayerNorm its own launch).  Measured inside the batch-32 pipeline on one box (profiles/r3_notes.md):
    Point-MAE only 23.55 ms per step against 23.67 unfolded; both 23.6-24.3; ViT only 24.3-24.9 -- hence the default."""
    v = os.environ.get("CMDIAD_LN_FOLD", "pmae")
    return v == "1" or (v not in ("0", "1") and v == net)


def _pack_block(sd, p, device, qkv_bias, net=""):
    g = lambda k: sd[p + k]  # noqa: E731
    blk = dict(
        ln1_w=_dev(g("norm1.weight"), device), ln1_b=_dev(g("norm1.bias"), device),
        ln2_w=_dev(g("norm2.weight"), device), ln2_b=_dev(g("norm2.bias"), device),
        qkv_w=_bf(g("attn.qkv.weight"), device),
        qkv_b=_dev(g("attn.qkv.bias"), device) if qkv_bias else None,
        proj_w=_bf(g("attn.proj.weight"), device), proj_b=_dev(g("attn.proj.bias"), device),
        fc1_w=_bf(g("mlp.fc1.weight"), device), fc1_b=_dev(g("mlp.fc1.bias"), device),
        fc2_w=_bf(g("mlp.fc2.weight"), device), fc2_b=_dev(g("mlp.fc2.bias"), device))
    if ln_fold_enabled(net):
        wq, bq = ln_fold(g("attn.qkv.weight"), g("attn.qkv.bias") if qkv_bias else None, g("norm1.weight"), g("norm1.bias"))
        w1, b1 = ln_fold(g("mlp.fc1.weight"), g("mlp.fc1.bias"), g("norm2.weight"), g("norm2.bias"))
        blk.update(qkv_wf=_bf(wq, device), qkv_bf=_dev(bq, device), fc1_wf=_bf(w1, device), fc1_bf=_dev(b1, device))
    return blk


def block_flags(i, n, folded, read_after=()):
    """LayerNorm-fold chaining flags of block i of n run back to back on one workspace: block i prepares block i + 1's first
    LayerNorm unless its output is read in between (read_after: Point-MAE's fetch layers) or it is the last."""
    if not folded:
        return 0
    prep = lambda j: 0 <= j < n - 1 and j not in read_after  # noqa: E731
    return (ops.BLOCK_LN1_READY if prep(i - 1) else 0) | (ops.BLOCK_PREP_NEXT if prep(i) else 0)


class _QkvBuffers:
    """q/k [B,H,Tp,64] and v^T [B,H,64,Tp]; padding rows are zeroed once and never written."""

    def __init__(self):
        self.key = None
        self.ws = None

    def workspace(self, M, C, hidden, device):
        need = ops.transformer_block_workspace_bytes(M, C, hidden)
        if self.ws is None or self.ws.numel() < need or self.ws.device != device:
            self.ws = torch.empty((need,), dtype=torch.uint8, device=device)
        return self.ws

    def get(self, B, H, T, device):
        key = (B, H, T, str(device))
        if self.key != key:
            Tp = (T + 63) // 64 * 64
            self.q = torch.zeros((B, H, Tp, 64), dtype=torch.bfloat16, device=device)
            self.k = torch.zeros_like(self.q)
            self.vt = torch.zeros((B, H, 64, Tp), dtype=torch.bfloat16, device=device)
            self.key = key
        return self.q, self.k, self.vt


def transformer_block(x, blk, B, T, H, eps, bufs, pos=None, flags=0):
    """In-place pre-LN block on the fp32 residual stream x [B*T, C] (models/models.py:177-180).
    pos (Point-MAE) is added to x first, fused into the first LayerNorm (models/models.py:240).
    One FFI call (cmdiad_transformer_block_fwd sequences the launches inside the library).  flags (block_flags): with the
    folded weights the LayerNorms run inside the products around them; PREP_NEXT / LN1_READY chain that across blocks."""
    q, k, vt = bufs.get(B, H, T, x.device)
    ops.transformer_block(x, pos, blk, B, T, H, eps, q, k, vt, bufs.workspace(B * T, x.shape[1], blk["fc1_w"].shape[0], x.device),
                          flags=flags)
    return x


def transformer_block_unfused(x, blk, B, T, H, eps, bufs, pos=None, flags=0, state=None):
    """The same block as separate entry-point calls (kept for tests: both forms must agree bit for bit).  state: a dict that
    carries the raw bf16 rows / 1 / sigma from a PREP_NEXT call to the LN1_READY call that follows it."""
    M, C = x.shape
    folded = "qkv_wf" in blk
    state = {} if state is None else state
    q, k, vt = bufs.get(B, H, T, x.device)
    if flags & ops.BLOCK_LN1_READY:
        ops.gemm_qkv(state["xb"], blk["qkv_wf"], blk["qkv_bf"], B, T, q, k, vt, row_scale=state["rstd"])
    else:
        h = ops.layernorm(x, blk["ln1_w"], blk["ln1_b"], eps, add=pos)
        ops.gemm_qkv(h, blk["qkv_w"], blk["qkv_b"], B, T, q, k, vt)
    a = ops.attention(q, k, vt, B, H, T)
    if folded:
        xb = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
        part = torch.empty((C // 64, M, 2), dtype=torch.float32, device=x.device)
        ops.gemm(a, blk["proj_w"], bias=blk["proj_b"], residual=x, out_f32=x, want_bf16=False, ln_xb=xb, ln_part=part)
        rstd = ops.ln_stats_finalize(part, M, C // 64, eps)
        _, m = ops.gemm(xb, blk["fc1_wf"], bias=blk["fc1_bf"], act=ops.ACT_GELU, row_scale=rstd)
    else:
        ops.gemm(a, blk["proj_w"], bias=blk["proj_b"], residual=x, out_f32=x, want_bf16=False)
        h = ops.layernorm(x, blk["ln2_w"], blk["ln2_b"], eps)
        _, m = ops.gemm(h, blk["fc1_w"], bias=blk["fc1_b"], act=ops.ACT_GELU)
    if flags & ops.BLOCK_PREP_NEXT:
        xb = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
        part = torch.empty((C // 64, M, 2), dtype=torch.float32, device=x.device)
        ops.gemm(m, blk["fc2_w"], bias=blk["fc2_b"], residual=x, out_f32=x, want_bf16=False, ln_xb=xb, ln_part=part, add2=pos)
        state["xb"], state["rstd"] = xb, ops.ln_stats_finalize(part, M, C // 64, eps)
    else:
        ops.gemm(m, blk["fc2_w"], bias=blk["fc2_b"], residual=x, out_f32=x, want_bf16=False)
    return x


# ------------------------------------------------------------------------------------------- ViT-B/8
class PackedViT:
    def __init__(self, sd, prefix="", device="cuda", depth=12, num_heads=12):
        self.device, self.depth, self.heads = device, depth, num_heads
        w = sd[prefix + "patch_embed.proj.weight"]
        self.dim = w.shape[0]
        self.patch_w = _bf(w.reshape(self.dim, -1), device)  # [768, 3*8*8], k = (c, dy, dx)
        self.patch_b = _dev(sd[prefix + "patch_embed.proj.bias"], device)
        self.cls = _dev(sd[prefix + "cls_token"].reshape(-1), device)
        self.pos = _dev(sd[prefix + "pos_embed"].reshape(-1, self.dim), device)
        self.blocks = [_pack_block(sd, f"{prefix}blocks.{i}.", device, True, "vit") for i in range(depth)]
        self.norm_w, self.norm_b = _dev(sd[prefix + "norm.weight"], device), _dev(sd[prefix + "norm.bias"], device)
        self.bufs = _QkvBuffers()

    def forward_tokens(self, rgb):
        """rgb [B,3,224,224] f32 cuda -> final-LayerNorm tokens [B, 785, 768] f32 (cls at index 0)."""
        B, _, S, _ = rgb.shape
        P = (S // 8) ** 2
        T = P + 1
        patches = ops.im2col_patch8(rgb.contiguous())
        po, _ = ops.gemm(patches, self.patch_w, bias=self.patch_b, want_f32=True, want_bf16=False)
        x = ops.vit_assemble(po, self.cls, self.pos, B, P, self.dim)
        n = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            transformer_block(x, blk, B, T, self.heads, 1e-6, self.bufs, flags=block_flags(i, n, "qkv_wf" in blk))
        out = torch.empty_like(x)
        ops.layernorm(x, self.norm_w, self.norm_b, 1e-6, out_f32=out, want_bf16=False)
        return out.view(B, T, self.dim)

    def forward(self, rgb):
        """-> [B,768,28,28] view, the reference's layout (models/models.py:52)."""
        tok = self.forward_tokens(rgb)
        B, T, C = tok.shape
        s = int((T - 1) ** 0.5)
        return tok[:, 1:].permute(0, 2, 1).reshape(B, C, s, s)


# ------------------------------------------------------------------------------------------- Point-MAE
def fold_pointmae_encoder(sd, prefix, device):
    """Eval-mode BatchNorm folded into the 1x1 convolutions (models/models.py:187-198)."""
    def bn(name):
        s = sd[prefix + name + ".weight"] / torch.sqrt(sd[prefix + name + ".running_var"] + 1e-5)
        return s, sd[prefix + name + ".bias"] - sd[prefix + name + ".running_mean"] * s

    s1, t1 = bn("first_conv.1")
    w1 = sd[prefix + "first_conv.0.weight"].reshape(128, 3) * s1[:, None]
    b1 = sd[prefix + "first_conv.0.bias"] * s1 + t1
    s2, t2 = bn("second_conv.1")
    w3 = sd[prefix + "second_conv.0.weight"].reshape(512, 512) * s2[:, None]
    b3 = sd[prefix + "second_conv.0.bias"] * s2 + t2
    w4 = sd[prefix + "second_conv.3.weight"]
    return dict(
        w1b1=_dev(torch.cat([w1, b1[:, None]], 1), device),
        W2=_bf(sd[prefix + "first_conv.3.weight"].reshape(256, 128), device),
        b2=_dev(sd[prefix + "first_conv.3.bias"], device),
        W3a=_bf(w3[:, :256].contiguous(), device),   # acts on the broadcast group maximum (cat order, :212)
        W3b=_bf(w3[:, 256:].contiguous(), device),   # acts on the per-point features
        b3=_dev(b3, device),
        W4=_bf(w4.reshape(w4.shape[0], 512), device), b4=_dev(sd[prefix + "second_conv.3.bias"], device))


def raw_pointmae_encoder(sd, prefix, device):
    """The encoder's parameters UNFOLDED (fp32, on the device) for the batch-statistics BatchNorm mode: the normalisation
    constants then depend on the sample (models/models.py:189,195 in training mode)."""
    g = lambda k: _dev(sd[prefix + k], device)  # noqa: E731
    return dict(w1=g("first_conv.0.weight").reshape(128, 3), b1=g("first_conv.0.bias"),
                g1=g("first_conv.1.weight"), be1=g("first_conv.1.bias"),
                w3=g("second_conv.0.weight").reshape(512, 512), b3=g("second_conv.0.bias"),
                g2=g("second_conv.1.weight"), be2=g("second_conv.1.bias"))


class PackedPointMAE:
    def __init__(self, sd, prefix="", device="cuda", depth=12, num_heads=6, taps=(3, 11), group_size=128, num_group=1024,
                 bn_batch_stats=None):
        """bn_batch_stats: True reproduces the reference AS SHIPPED -- its extractor is never put in .eval(), so the two
        BatchNorm1d layers of the encoder normalise with the statistics of the sample at hand (DropPath, also
        active there, is RNG-dependent and stays off).  Default (None): the CMDIAD_BN_BATCH_STATS=1 environment switch, else
        the eval-mode contract of record (running statistics folded into the convolutions)."""
        self.device, self.depth, self.heads, self.taps = device, depth, num_heads, taps
        self.group_size, self.num_group = group_size, num_group
        self.bn_batch_stats = (os.environ.get("CMDIAD_BN_BATCH_STATS", "0") == "1") if bn_batch_stats is None else bool(bn_batch_stats)
        self.enc = fold_pointmae_encoder(sd, prefix + "encoder.", device)
        self.enc_raw = raw_pointmae_encoder(sd, prefix + "encoder.", device) if self.bn_batch_stats else None
        self.dim = self.enc["W4"].shape[0]
        self.pos0 = _dev(torch.cat([sd[prefix + "pos_embed.0.weight"], sd[prefix + "pos_embed.0.bias"][:, None]], 1), device)
        self.pos2_w = _bf(sd[prefix + "pos_embed.2.weight"], device)
        self.pos2_b = _dev(sd[prefix + "pos_embed.2.bias"], device)
        self.blocks = [_pack_block(sd, f"{prefix}blocks.blocks.{i}.", device, False, "pmae") for i in range(depth)]
        self.norm_w, self.norm_b = _dev(sd[prefix + "norm.weight"], device), _dev(sd[prefix + "norm.bias"], device)
        self.bufs = _QkvBuffers()

    def encode_batch_stats(self, neighborhood, eps=1e-5):
        """models/models.py:200-215 with BOTH BatchNorm1d layers in training mode: every sample is normalised with its own
        (biased) batch statistics -- the reference drives the extractor at batch size 1, so 'the batch' is the sample's
        G x Mg points.  Same kernels as the eval path; the folded constants are rebuilt per sample:
          BN1: conv1 is linear in the coordinates, so mean / variance of its 128 outputs follow from the mean vector and
               covariance matrix of the sample's points (cmdiad_moments3): mean_c = w_c . mu + b_c, var_c = w_c^T S w_c;
          BN2: a statistics pass evaluates conv3's pre-activation z = W3b . h2 + (W3a . gmax + b3) in fp32 and reduces its
               column moments (cmdiad_col_moments); then scale = gamma / sqrt(var + eps) goes into W3b and the group bias."""
        B, G, Mg, _ = neighborhood.shape
        e, r = self.enc, self.enc_raw
        toks = []
        for b in range(B):
            pts = neighborhood[b].reshape(-1, 3).contiguous()
            mu, cov = ops.moments3(pts)
            w1 = r["w1"].double()
            mean1 = w1 @ mu + r["b1"].double()
            var1 = ((w1 @ cov) * w1).sum(1).clamp_min(0.0)
            s1 = r["g1"].double() / torch.sqrt(var1 + eps)
            w1b1 = torch.cat([w1 * s1[:, None], ((r["b1"].double() - mean1) * s1 + r["be1"].double())[:, None]], 1).float().contiguous()
            h2, _, g16 = ops.encoder_stage1(pts, w1b1, e["W2"], e["b2"], G, Mg)
            gb_raw, _ = ops.gemm(g16, e["W3a_raw"], bias=r["b3"], want_f32=True, want_bf16=False)
            z, _ = ops.gemm(h2, e["W3b_raw"], group_bias=gb_raw, group_rows=Mg, want_f32=True, want_bf16=False)
            mean2, var2 = ops.col_moments(z)
            del z
            s2 = r["g2"].double() / torch.sqrt(var2.clamp_min(0.0) + eps)
            w3b = ops.cast_bf16((r["w3"][:, 256:].double() * s2[:, None]).float().contiguous())
            gb = ((gb_raw.double() - mean2) * s2 + r["be2"].double()).float().contiguous()
            toks.append(ops.encoder_tail(h2, gb, w3b, e["W4"], e["b4"], G, Mg))
        return torch.cat(toks, 0)

    def encode(self, neighborhood):
        """neighborhood [B,G,Mg,3] f32 -> tokens [B*G, 384] f32 (models/models.py:200-215)."""
        B, G, Mg, _ = neighborhood.shape
        e = self.enc
        if self.bn_batch_stats:
            if "W3a_raw" not in e:
                e["W3a_raw"] = ops.cast_bf16(self.enc_raw["w3"][:, :256].contiguous())
                e["W3b_raw"] = ops.cast_bf16(self.enc_raw["w3"][:, 256:].contiguous())
            return self.encode_batch_stats(neighborhood)
        h2, _, g16 = ops.encoder_stage1(neighborhood.reshape(-1, 3), e["w1b1"], e["W2"], e["b2"], B * G, Mg)
        gb, _ = ops.gemm(g16, e["W3a"], bias=e["b3"], want_f32=True, want_bf16=False)
        if os.environ.get("CMDIAD_ENCODER_TAIL", "1") == "1":
            # conv3 (per-point half) + ReLU + conv4 + group max in one kernel: h3 (4.3 GB at batch 32) never leaves LDS
            return ops.encoder_tail(h2, gb, e["W3b"], e["W4"], e["b4"], B * G, Mg)
        _, h3 = ops.gemm(h2, e["W3b"], act=ops.ACT_RELU, group_bias=gb, group_rows=Mg)
        tok, _ = ops.gemm_groupmax(h3, e["W4"], e["b4"], B * G, Mg)
        return tok

    def transform(self, tokens, center):
        """tokens [B*G,384] f32 (consumed), center [B,G,3] -> feats [B, G, 768] f32, centre-major
        (the reference's [B,768,G] is ``feats.transpose(1, 2)``; models/models.py:234-243, 360-373)."""
        B, G, _ = center.shape
        C = self.dim
        p1 = ops.linear3(center.reshape(-1, 3), self.pos0, ops.ACT_GELU)
        pos, _ = ops.gemm(p1, self.pos2_w, bias=self.pos2_b, want_f32=True, want_bf16=False)
        feats = torch.empty((B * G, C * len(self.taps)), dtype=torch.float32, device=tokens.device)
        x = tokens
        t = 0
        n = len(self.blocks)
        for i, blk in enumerate(self.blocks):
            # (a fetch layer's output is read before the next block adds pos to it: that block runs its own first LayerNorm)
            transformer_block(x, blk, B, G, self.heads, 1e-5, self.bufs, pos=pos, flags=block_flags(i, n, "qkv_wf" in blk, self.taps))
            if i in self.taps:
                ops.layernorm(x, self.norm_w, self.norm_b, 1e-5, out_f32=feats[:, t * C:(t + 1) * C], want_bf16=False)
                t += 1
        return feats.view(B, G, -1)

    def sample(self, xyz, n_valid=None):
        """Farthest-point sampling alone (models/models.py:70-78) -> (center_idx [B,G] int32, center [B,G,3]): a caller that
        launches eagerly queues this first -- 1 023 dependent rounds on one CU per cloud -- and everything else under it."""
        return ops.fps(xyz, self.num_group, n_valid)

    def forward(self, xyz, n_valid=None, sampled=None):
        """xyz [B,N,3] f32 cuda (rows >= n_valid[b] are padding) ->
        (feats [B,G,768] centre-major, center [B,G,3], ori_idx [B,G,Mg] int64, center_idx [B,G] int32).
        sampled: the result of sample() on the same cloud, when it was queued earlier."""
        center_idx, center = sampled if sampled is not None else ops.fps(xyz, self.num_group, n_valid)
        ori_idx, nb = ops.knn_group(xyz, center, self.group_size, n_valid)
        tok = self.encode(nb)
        feats = self.transform(tok, center)
        return feats, center, ori_idx, center_idx


# ------------------------------------------------------------------------------------------- hallucination MLP
class PackedHallucination:
    """Inference-side packing of HallucinationCrossModalityNetwork (models/hallucination_network.py:18-45); mlp_depth
    chained MlpBlocks per direction (utils/utils.py:103-115), one in the reference's default."""

    def __init__(self, sd, device="cuda"):
        self.dir = {}
        for name in ("xyz", "rgb"):
            blocks = []
            d = 0
            while f"{name}_mlp.mlp_module.{d}.fc1.weight" in sd:
                p = f"{name}_mlp.mlp_module.{d}."
                blocks.append(dict(w1=_bf(sd[p + "fc1.weight"], device), b1=_dev(sd[p + "fc1.bias"], device),
                                   w2=_bf(sd[p + "fc2.weight"], device), b2=_dev(sd[p + "fc2.bias"], device),
                                   w3=_bf(sd[p + "fc3.weight"], device), b3=_dev(sd[p + "fc3.bias"], device)))
                d += 1
            if not blocks:
                raise KeyError(f"{name}_mlp.mlp_module.0.fc1.weight: not a HallucinationCrossModalityNetwork state_dict")
            self.dir[name] = dict(ln_w=_dev(sd[f"{name}_norm.weight"], device), ln_b=_dev(sd[f"{name}_norm.bias"], device),
                                  blocks=blocks, **blocks[0])

    def generate(self, x, src, m_count=None):
        """src='xyz': xyz features -> hallucinated rgb features (out_type='rgb'); src='rgb': the reverse.
        x [..., D] f32 cuda -> same leading shape, f32.  LN -> (fc1 -> GELU -> fc2 -> GELU -> fc3 -> GELU) x mlp_depth.
        m_count (device int32 [1]): x is a compacted row set (cmdiad_rows_dedup_plan) of which only the first m_count rows are
        live -- the products run on those rows only (every output row depends on its own input row alone), the rest of the
        output is left unwritten."""
        w = self.dir[src]
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        h = ops.layernorm(x2, w["ln_w"], w["ln_b"], 1e-5)
        for d, b in enumerate(w["blocks"]):
            last = d == len(w["blocks"]) - 1
            _, h = ops.gemm(h, b["w1"], bias=b["b1"], act=ops.ACT_GELU, m_count=m_count)
            _, h = ops.gemm(h, b["w2"], bias=b["b2"], act=ops.ACT_GELU, m_count=m_count)
            out, h = ops.gemm(h, b["w3"], bias=b["b3"], act=ops.ACT_GELU, want_f32=last, want_bf16=not last, m_count=m_count)
        return out.view(*shape[:-1], out.shape[-1])


# ------------------------------------------------------------------------------------------- convolution heads (8f/f4)
def _fold_conv_bn(sd, conv, bn, device, cin_pad=None, n_pad=None, eps=1e-5):
    """Conv2d weight [N,C,k,k] (+ optional bias) with an eval-mode BatchNorm2d folded in -> (W [N', k*k*C'] bf16 tap-major,
    bias [N'] f32 or None).  Zero rows / channels pad N and C up to n_pad / cin_pad (a zero output column stays zero
    through ReLU, so padded activations feed the next layer's padded channels)."""
    w = sd[conv + ".weight"].detach().float()
    b = sd[conv + ".bias"].detach().float() if (conv + ".bias") in sd else None
    if bn is not None:
        g = sd[bn + ".weight"].detach().float() / torch.sqrt(sd[bn + ".running_var"].detach().float() + eps)
        w = w * g.view(-1, 1, 1, 1)
        b = sd[bn + ".bias"].detach().float() - sd[bn + ".running_mean"].detach().float() * g + (b * g if b is not None else 0)
    N, C, kh, kw = w.shape
    Np, Cp = n_pad or N, cin_pad or C
    wp = torch.zeros((Np, kh, kw, Cp), dtype=torch.float32)
    wp[:N, :, :, :C] = w.permute(0, 2, 3, 1).cpu()
    bias = None
    if b is not None:
        bias = torch.zeros(Np, dtype=torch.float32)
        bias[:N] = b.cpu()
        bias = bias.to(device)
    return wp.reshape(Np, -1).to(device).to(torch.bfloat16).contiguous(), bias


def _tokens_as_image(x, device):
    """[B, HW, C] f32 tokens -> bf16 NHWC image [B, H, W, C] (hallucination_network.py:6-9 is only a reshape)."""
    B, T, C = x.shape
    side = int(round(T ** 0.5))
    if side * side != T:
        raise ValueError(f"token count {T} is not a square feature map")
    return ops.cast_bf16(x.to(device, torch.float32).contiguous()).view(B, side, side, C)


class PackedConvFtoF:
    """HallucinationCrossModalityConv (models/hallucination_network.py:72-143), eval mode: per direction
    [conv3x3 + BN + ReLU] x 3 + conv3x3, 768 channels on the 56 x 56 token map, as four implicit-GEMM launches."""

    def __init__(self, sd, device="cuda"):
        self.device = device
        self.dir = {}
        for name in ("xyz", "rgb"):
            layers = []
            for i in range(4):
                bn = f"{name}_conv.{3 * i + 1}" if i < 3 else None
                layers.append(_fold_conv_bn(sd, f"{name}_conv.{3 * i}", bn, device))
            self.dir[name] = layers

    def generate(self, x, src):
        """src='xyz': xyz features -> hallucinated rgb features (xyz_conv); src='rgb': the reverse.  [B,3136,768] -> same."""
        h = _tokens_as_image(x, self.device)
        B, H, W, _ = h.shape
        for i, (w, b) in enumerate(self.dir[src]):
            last = i == 3
            o32, o16 = ops.conv2d_nhwc(h, w, w.shape[0], 3, 1, bias=b, act=ops.ACT_NONE if last else ops.ACT_RELU,
                                       want_f32=last, want_bf16=not last)
            h = o32 if last else o16
        return h.view(B, H * W, h.shape[-1])


class PackedFtoIConv:
    """HallucinationFeatureToInputConv (models/hallucination_network.py:185-220): conv 768 -> 384 on the 56 x 56 map, bicubic
    to 224 x 224, conv 384 -> 96 (ReLU) -> 32 (ReLU) -> 3.  Channel counts that are not multiples of 64 are zero-padded."""

    def __init__(self, sd, device="cuda"):
        self.device = device
        self.c1 = _fold_conv_bn(sd, "conv1", None, device)
        self.c2 = _fold_conv_bn(sd, "conv2", None, device, n_pad=128)
        self.c3 = _fold_conv_bn(sd, "conv3", None, device, cin_pad=128, n_pad=64)
        self.c4 = _fold_conv_bn(sd, "conv4", None, device, cin_pad=64, n_pad=4)
        self.out_dim = sd["conv4.weight"].shape[0]

    def generate(self, feature):
        h = _tokens_as_image(feature, self.device)
        B = h.shape[0]
        f32, _ = ops.conv2d_nhwc(h, self.c1[0], 384, 3, 1, bias=self.c1[1], want_f32=True, want_bf16=False)
        h = ops.upsample_bicubic(f32, 384, 224, 224)
        _, h = ops.conv2d_nhwc(h, self.c2[0], 128, 3, 1, bias=self.c2[1], act=ops.ACT_RELU)
        _, h = ops.conv2d_nhwc(h, self.c3[0], 64, 3, 1, bias=self.c3[1], act=ops.ACT_RELU)
        out, _ = ops.conv2d_nhwc(h, self.c4[0], 4, 3, 1, bias=self.c4[1], want_f32=True, want_bf16=False)
        return out.view(B, 224, 224, 4)[..., :self.out_dim].permute(0, 3, 1, 2).contiguous()


class PackedFtoIMLP:
    """HallucinationRGBFeatureToXYZInputMLP (models/hallucination_network.py:146-182): LayerNorm, 768 -> 1152 -> 384 -> 96 ->
    out_dim with GELU between, bicubic 56 -> 224.  The 96-wide layer is zero-padded to 128 (K % 64 == 0), out_dim to 4."""

    def __init__(self, sd, device="cuda"):
        self.device = device
        self.ln = (_dev(sd["rgb_norm.weight"], device), _dev(sd["rgb_norm.bias"], device))
        self.out_dim = sd["mlp.6.weight"].shape[0]

        def lin(i, n_pad=None, k_pad=None):
            w, b = sd[f"mlp.{i}.weight"].detach().float().cpu(), sd[f"mlp.{i}.bias"].detach().float().cpu()
            wp = torch.zeros((n_pad or w.shape[0], k_pad or w.shape[1]))
            wp[:w.shape[0], :w.shape[1]] = w
            bp = torch.zeros(wp.shape[0])
            bp[:b.shape[0]] = b
            return wp.to(device).to(torch.bfloat16).contiguous(), bp.to(device)

        self.l = [lin(0), lin(2), lin(4, n_pad=128), lin(6, n_pad=4, k_pad=128)]

    def generate(self, x):
        B, T, C = x.shape
        side = int(round(T ** 0.5))
        h = ops.layernorm(x.to(self.device, torch.float32).reshape(B * T, C).contiguous().clone(), self.ln[0], self.ln[1], 1e-5)
        for w, b in self.l[:3]:
            _, h = ops.gemm(h, w, bias=b, act=ops.ACT_GELU)
        out, _ = ops.gemm(h, self.l[3][0], bias=self.l[3][1], want_f32=True, want_bf16=False)
        return ops.upsample_bicubic(out.view(B, side, side, 4), self.out_dim, 224, 224, nchw=True)


class PackedHRNet:
    """The trunk models/hrnet.py actually runs (hrnet.py:251-288): stem 3 -> 64 -> 128 (3x3, stride 2, BN, ReLU), twelve
    Bottlenecks 512 -> 128 -> 128 -> 512 (layer1-3; the first one widens 128 -> 512 through its `downsample` branch), 1x1
    final_layer.  The residual trunk stays fp32, every convolution operand is bf16."""

    def __init__(self, sd, device="cuda"):
        self.device = device
        g = sd["bn1.weight"].detach().float() / torch.sqrt(sd["bn1.running_var"].detach().float() + 1e-5)
        self.stem_w = (sd["conv1.weight"].detach().float() * g.view(-1, 1, 1, 1)).to(device).contiguous()
        self.stem_b = (sd["bn1.bias"].detach().float() - sd["bn1.running_mean"].detach().float() * g).to(device).contiguous()
        self.stem2 = _fold_conv_bn(sd, "conv2", "bn2", device)
        self.blocks = []
        for layer in (1, 2, 3):
            for i in range(4):
                p = f"layer{layer}.{i}."
                blk = [_fold_conv_bn(sd, p + f"conv{k}", p + f"bn{k}", device) for k in (1, 2, 3)]
                down = _fold_conv_bn(sd, p + "downsample.0", p + "downsample.1", device) if (p + "downsample.0.weight") in sd else None
                self.blocks.append((blk, down))
        self.final = _fold_conv_bn(sd, "final_layer", None, device)
        if self.final[0].shape[1] != 512:
            raise ValueError(f"HRNet(c={self.final[0].shape[1]}): final_layer expects c channels but the trunk is 512 wide "
                             "(hrnet.py:248,286) -- construct it with c=512 (--c_hrnet 512)")

    def generate(self, img):
        """[B,3,H,W] f32 -> tokens [B, (H/4)*(W/4), nof_joints] f32."""
        x = ops.conv_stem(img.to(self.device, torch.float32).contiguous(), self.stem_w, self.stem_b, 2)
        _, x16 = ops.conv2d_nhwc(x, self.stem2[0], 128, 3, 2, bias=self.stem2[1], act=ops.ACT_RELU)
        x32 = None
        for (c1, c2, c3), down in self.blocks:
            if down is not None:
                x32, _ = ops.conv2d_nhwc(x16, down[0], down[0].shape[0], 1, 1, bias=down[1], want_f32=True, want_bf16=False)
            _, t = ops.conv2d_nhwc(x16, c1[0], c1[0].shape[0], 1, 1, bias=c1[1], act=ops.ACT_RELU)
            _, t = ops.conv2d_nhwc(t, c2[0], c2[0].shape[0], 3, 1, bias=c2[1], act=ops.ACT_RELU)
            x32, x16 = ops.conv2d_nhwc(t, c3[0], c3[0].shape[0], 1, 1, bias=c3[1], act=ops.ACT_RELU_POST, residual=x32,
                                       want_f32=True, want_bf16=True)
        out, _ = ops.conv2d_nhwc(x16, self.final[0], self.final[0].shape[0], 1, 1, bias=self.final[1], want_f32=True, want_bf16=False)
        B, H, W, N = out.shape
        return out.view(B, H * W, N)
