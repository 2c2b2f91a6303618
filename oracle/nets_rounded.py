"""Operand-rounded fp64 restatement of the two transformers' forward passes: the algebra of ``oracle/nets.py`` with every
matrix-product operand rounded to bf16 at the places where the HIP path rounds it, everything else in float64.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/ alone.

Why: against the plain fp32 oracle the bf16 chain of 12 blocks differs by ~1 % of the feature scale, so those tests
(tests/test_gpu_nets.py: mean 1.5 %, max 12 %) cannot see a small systematic error -- a wrong bias on one output column, a
mis-scaled head.  With the SAME operand roundings on both sides what is left is accumulation order (fp32 vs fp64), the exp2 /
GELU approximations (<= 1e-6) and a handful of bf16 roundings that flip on a last-bit difference: three orders of magnitude
tighter (tests assert max |err| <= 3e-3 of the feature scale).

Rounding points of the HIP path (cmdiad_amd/runtime.py, csrc/gemm.hip, csrc/attention.hip), restated here:
  * weights of every product: bf16 (runtime._bf); the first point-cloud convolution and all biases stay fp32;
  * LayerNorm output -> bf16 (the A operand of qkv / fc1); final LayerNorms -> fp32;
  * q = bf16((x.Wq + b) * head_dim^-0.5 * log2 e), k, v = bf16(x.W + b)   (csrc/gemm.hip gemm_qkv_kernel);
  * attention: scores in the log2 domain, keys in tiles of 64 with a running maximum; P = bf16(exp2(S - m_running)) is the
    operand of P.V, the row sum uses the unrounded exp2 values; output bf16(O / l)                 (csrc/attention.hip);
  * fc1: bf16(gelu(acc + bias)); proj / fc2: fp32 acc + bias + residual (the fp32 residual stream);
  * ViT patches: bf16(im2col(rgb)); Point-MAE: h1 = bf16(relu(conv1)) (conv1 in fp32 with the BatchNorm folded in fp32),
    h2 = bf16(conv2), group maxima of the bf16 values, h3 = bf16(relu(conv3)), tokens fp32; pos = fc(bf16(gelu(fc(centre)))).
Reference anchors as in oracle/nets.py (models/models.py:35-53, 126-215, 218-243, 352-373).
"""
import math

import torch
import torch.nn.functional as F

LOG2E = 1.4426950408889634


def r16(x):
    """round-to-nearest-even to bf16 (from the fp32 value the GPU holds), back in float64"""
    return x.float().to(torch.bfloat16).double()


def _lin(x, w, b=None):
    y = x @ r16(w).T
    return y if b is None else y + b.double()


def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double(), eps)


def _attention(h, sd, prefix, num_heads, tile=64):
    """h: bf16-valued LayerNorm output [B,T,C] (float64).  csrc/gemm.hip gemm_qkv_kernel + csrc/attention.hip."""
    B, T, C = h.shape
    hd = C // num_heads
    qkv = _lin(h, sd[prefix + ".qkv.weight"], sd.get(prefix + ".qkv.bias")).reshape(B, T, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    scale = torch.tensor(hd ** -0.5 * LOG2E, dtype=torch.float32)
    q = r16(qkv[0].float() * scale)                  # the scale is applied to the fp32 value, then rounded
    k, v = r16(qkv[1]), r16(qkv[2])
    s = q @ k.transpose(-2, -1)                      # [B,H,T,T], log2 domain
    m = torch.full((B, num_heads, T), -math.inf, dtype=torch.float64)
    l = torch.zeros((B, num_heads, T), dtype=torch.float64)
    o = torch.zeros((B, num_heads, T, hd), dtype=torch.float64)
    for t0 in range(0, T, tile):
        st = s[..., t0:t0 + tile]
        m_new = torch.maximum(m, st.amax(-1))
        alpha = torch.exp2(m - m_new)                # 0 for the first tile (m = -inf)
        p = torch.exp2(st - m_new[..., None])
        l = l * alpha + p.sum(-1)
        o = o * alpha[..., None] + r16(p) @ v[:, :, t0:t0 + tile]
        m = m_new
    a = r16(o / l[..., None]).transpose(1, 2).reshape(B, T, C)
    return _lin(a, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"])


def _block(x, sd, prefix, num_heads, eps):
    x = x + _attention(r16(_ln(x, sd, prefix + ".norm1", eps)), sd, prefix + ".attn", num_heads)
    h = r16(F.gelu(_lin(r16(_ln(x, sd, prefix + ".norm2", eps)), sd[prefix + ".mlp.fc1.weight"], sd[prefix + ".mlp.fc1.bias"])))
    return x + _lin(h, sd[prefix + ".mlp.fc2.weight"], sd[prefix + ".mlp.fc2.bias"])


def vit_forward_rounded(sd, rgb, prefix="", depth=12, num_heads=12, patch=8, eps=1e-6):
    """rgb [B,3,224,224] -> [B,768,28,28] float64 (models/models.py:41-52; runtime.PackedViT.forward)."""
    x = F.conv2d(r16(rgb), r16(sd[prefix + "patch_embed.proj.weight"]), sd[prefix + "patch_embed.proj.bias"].double(), stride=patch)
    B, C, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[prefix + "cls_token"].double().expand(B, -1, -1), x], dim=1) + sd[prefix + "pos_embed"].double()
    for i in range(depth):
        x = _block(x, sd, f"{prefix}blocks.{i}", num_heads, eps)
    x = _ln(x, sd, prefix + "norm", eps)
    return x[:, 1:].permute(0, 2, 1).reshape(B, C, gh, gw)


def pointmae_encoder_rounded(sd, neighborhood, prefix="encoder."):
    """neighborhood [B,G,M,3] -> tokens [B,G,384] float64 (models/models.py:200-215, eval-mode BatchNorm folded as
    runtime.fold_pointmae_encoder does: in fp32, before the bf16 cast of the products' weights)."""
    def bn(name):
        s = sd[prefix + name + ".weight"] / torch.sqrt(sd[prefix + name + ".running_var"] + 1e-5)
        return s, sd[prefix + name + ".bias"] - sd[prefix + name + ".running_mean"] * s
    s1, t1 = bn("first_conv.1")
    w1 = (sd[prefix + "first_conv.0.weight"].reshape(128, 3) * s1[:, None]).double()      # conv1 runs in fp32: not rounded
    b1 = (sd[prefix + "first_conv.0.bias"] * s1 + t1).double()
    s2, t2 = bn("second_conv.1")
    w3 = sd[prefix + "second_conv.0.weight"].reshape(512, 512) * s2[:, None]
    b3 = (sd[prefix + "second_conv.0.bias"] * s2 + t2).double()
    x = neighborhood.double()
    h1 = r16(F.relu(x @ w1.T + b1))
    h2 = r16(_lin(h1, sd[prefix + "first_conv.3.weight"].reshape(256, 128), sd[prefix + "first_conv.3.bias"]))
    g = h2.amax(dim=2)                                                                    # [B,G,256], bf16 values
    gb = g @ r16(w3[:, :256]).T + b3                                                      # fp32 on the GPU
    h3 = r16(F.relu(h2 @ r16(w3[:, 256:]).T + gb[:, :, None, :]))
    w4 = sd[prefix + "second_conv.3.weight"]
    return _lin(h3, w4.reshape(w4.shape[0], 512), sd[prefix + "second_conv.3.bias"]).amax(dim=2)


def pointmae_transformer_rounded(sd, tokens, center, prefix="", depth=12, num_heads=6, taps=(3, 11), eps=1e-5):
    """tokens [B,G,384], center [B,G,3] -> [B,768,G] float64 (models/models.py:234-243, 360-373; PackedPointMAE.transform with
    the LayerNorms as launches, CMDIAD_LN_FOLD=0)."""
    p1 = r16(F.gelu(center.double() @ sd[prefix + "pos_embed.0.weight"].double().T + sd[prefix + "pos_embed.0.bias"].double()))
    pos = _lin(p1, sd[prefix + "pos_embed.2.weight"], sd[prefix + "pos_embed.2.bias"])
    x = tokens.double()
    outs = []
    for i in range(depth):
        x = _block(x + pos, sd, f"{prefix}blocks.blocks.{i}", num_heads, eps)             # pos re-added every layer
        if i in taps:
            outs.append(_ln(x, sd, prefix + "norm", eps).transpose(-1, -2))
    return torch.cat(outs, dim=1)
