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


def _pack_block(sd, p, device, qkv_bias):
    g = lambda k: sd[p + k]  # noqa: E731
    return dict(
        ln1_w=_dev(g("norm1.weight"), device), ln1_b=_dev(g("norm1.bias"), device),
        ln2_w=_dev(g("norm2.weight"), device), ln2_b=_dev(g("norm2.bias"), device),
        qkv_w=_bf(g("attn.qkv.weight"), device),
        qkv_b=_dev(g("attn.qkv.bias"), device) if qkv_bias else None,
        proj_w=_bf(g("attn.proj.weight"), device), proj_b=_dev(g("attn.proj.bias"), device),
        fc1_w=_bf(g("mlp.fc1.weight"), device), fc1_b=_dev(g("mlp.fc1.bias"), device),
        fc2_w=_bf(g("mlp.fc2.weight"), device), fc2_b=_dev(g("mlp.fc2.bias"), device))


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


def transformer_block(x, blk, B, T, H, eps, bufs, pos=None):
    """In-place pre-LN block on the fp32 residual stream x [B*T, C] (models/models.py:177-180).
    pos (Point-MAE) is added to x first, fused into the first LayerNorm (models/models.py:240).
    One FFI call (cmdiad_transformer_block_fwd sequences the seven launches inside the library)."""
    q, k, vt = bufs.get(B, H, T, x.device)
    ops.transformer_block(x, pos, blk, B, T, H, eps, q, k, vt, bufs.workspace(B * T, x.shape[1], blk["fc1_w"].shape[0], x.device))
    return x


def transformer_block_unfused(x, blk, B, T, H, eps, bufs, pos=None):
    """The same block as seven separate entry-point calls (kept for tests: both forms must agree bit for bit)."""
    h = ops.layernorm(x, blk["ln1_w"], blk["ln1_b"], eps, add=pos)
    q, k, vt = bufs.get(B, H, T, x.device)
    ops.gemm_qkv(h, blk["qkv_w"], blk["qkv_b"], B, T, q, k, vt)
    a = ops.attention(q, k, vt, B, H, T)
    ops.gemm(a, blk["proj_w"], bias=blk["proj_b"], residual=x, out_f32=x, want_bf16=False)
    h = ops.layernorm(x, blk["ln2_w"], blk["ln2_b"], eps)
    _, m = ops.gemm(h, blk["fc1_w"], bias=blk["fc1_b"], act=ops.ACT_GELU)
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
        self.blocks = [_pack_block(sd, f"{prefix}blocks.{i}.", device, True) for i in range(depth)]
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
        for blk in self.blocks:
            transformer_block(x, blk, B, T, self.heads, 1e-6, self.bufs)
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


class PackedPointMAE:
    def __init__(self, sd, prefix="", device="cuda", depth=12, num_heads=6, taps=(3, 11), group_size=128, num_group=1024):
        self.device, self.depth, self.heads, self.taps = device, depth, num_heads, taps
        self.group_size, self.num_group = group_size, num_group
        self.enc = fold_pointmae_encoder(sd, prefix + "encoder.", device)
        self.dim = self.enc["W4"].shape[0]
        self.pos0 = _dev(torch.cat([sd[prefix + "pos_embed.0.weight"], sd[prefix + "pos_embed.0.bias"][:, None]], 1), device)
        self.pos2_w = _bf(sd[prefix + "pos_embed.2.weight"], device)
        self.pos2_b = _dev(sd[prefix + "pos_embed.2.bias"], device)
        self.blocks = [_pack_block(sd, f"{prefix}blocks.blocks.{i}.", device, False) for i in range(depth)]
        self.norm_w, self.norm_b = _dev(sd[prefix + "norm.weight"], device), _dev(sd[prefix + "norm.bias"], device)
        self.bufs = _QkvBuffers()

    def encode(self, neighborhood):
        """neighborhood [B,G,Mg,3] f32 -> tokens [B*G, 384] f32 (models/models.py:200-215)."""
        B, G, Mg, _ = neighborhood.shape
        e = self.enc
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
        for i, blk in enumerate(self.blocks):
            transformer_block(x, blk, B, G, self.heads, 1e-5, self.bufs, pos=pos)
            if i in self.taps:
                ops.layernorm(x, self.norm_w, self.norm_b, 1e-5, out_f32=feats[:, t * C:(t + 1) * C], want_bf16=False)
                t += 1
        return feats.view(B, G, -1)

    def forward(self, xyz, n_valid=None):
        """xyz [B,N,3] f32 cuda (rows >= n_valid[b] are padding) ->
        (feats [B,G,768] centre-major, center [B,G,3], ori_idx [B,G,Mg] int64, center_idx [B,G] int32)."""
        center_idx, center = ops.fps(xyz, self.num_group, n_valid)
        ori_idx, nb = ops.knn_group(xyz, center, self.group_size, n_valid)
        tok = self.encode(nb)
        feats = self.transform(tok, center)
        return feats, center, ori_idx, center_idx


# ------------------------------------------------------------------------------------------- hallucination MLP
class PackedHallucination:
    """Inference-side packing of HallucinationCrossModalityNetwork (models/hallucination_network.py:18-45)."""

    def __init__(self, sd, device="cuda"):
        self.dir = {}
        for name in ("xyz", "rgb"):
            p = f"{name}_mlp.mlp_module.0."
            self.dir[name] = dict(
                ln_w=_dev(sd[f"{name}_norm.weight"], device), ln_b=_dev(sd[f"{name}_norm.bias"], device),
                w1=_bf(sd[p + "fc1.weight"], device), b1=_dev(sd[p + "fc1.bias"], device),
                w2=_bf(sd[p + "fc2.weight"], device), b2=_dev(sd[p + "fc2.bias"], device),
                w3=_bf(sd[p + "fc3.weight"], device), b3=_dev(sd[p + "fc3.bias"], device))

    def generate(self, x, src):
        """src='xyz': xyz features -> hallucinated rgb features (out_type='rgb'); src='rgb': the reverse.
        x [..., D] f32 cuda -> same leading shape, f32.  LN -> fc1 -> GELU -> fc2 -> GELU -> fc3 -> GELU."""
        w = self.dir[src]
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).contiguous()
        h = ops.layernorm(x2, w["ln_w"], w["ln_b"], 1e-5)
        _, h = ops.gemm(h, w["w1"], bias=w["b1"], act=ops.ACT_GELU)
        _, h = ops.gemm(h, w["w2"], bias=w["b2"], act=ops.ACT_GELU)
        out, _ = ops.gemm(h, w["w3"], bias=w["b3"], act=ops.ACT_GELU, want_f32=True, want_bf16=False)
        return out.view(*shape[:-1], out.shape[-1])
