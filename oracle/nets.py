"""Torch-CPU fp32 restatement of the floating-point networks on CMDIAD's hot path.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by cmdiad_amd/.

Everything here is a pure function over a ``state_dict`` whose keys are the
reference's own parameter names, so real checkpoints and the seeded synthetic
weights of tests/golden/make_golden.py load unchanged.

Reference anchors (file:line into evenrose/CMDIAD):
  * ViT-B/8 forward ............ models/models.py:35-53; the arithmetic is timm==0.9.12
    ``VisionTransformer`` (requirements.txt:12), NOT vendored -> restated from its
    published definition: conv patch-embed 8x8/8, cls token + learned pos-embed (785),
    12 pre-LN blocks (LN eps 1e-6, qkv bias, 12 heads x 64, erf-GELU MLP x4), final LN.
    "parity unpinned" for patch-embed/pos-embed; the block algebra is pinned through the
    reference's in-tree ``Block`` (models/models.py:163-180), which is the same algebra.
  * Point-MAE encoder .......... models/models.py:183-215
  * Point-MAE transformer ...... models/models.py:135-180, 218-243, 268-282, 352-373
  * hallucination MLP + losses . models/hallucination_network.py:18-69, utils/utils.py:86-115
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _linear(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _attention(x, sd, prefix, num_heads):
    # models/models.py:148-160 (and timm's Attention, same algebra)
    B, T, C = x.shape
    hd = C // num_heads
    qkv = _linear(x, sd, prefix + ".qkv").reshape(B, T, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    y = (att @ v).transpose(1, 2).reshape(B, T, C)
    return _linear(y, sd, prefix + ".proj")


def _block(x, sd, prefix, num_heads, eps):
    # models/models.py:177-180 with DropPath inactive (eval contract)
    x = x + _attention(_ln(x, sd, prefix + ".norm1", eps), sd, prefix + ".attn", num_heads)
    h = _linear(_ln(x, sd, prefix + ".norm2", eps), sd, prefix + ".mlp.fc1")
    h = _linear(F.gelu(h), sd, prefix + ".mlp.fc2")
    return x + h


# ----------------------------------------------------------------------------- ViT-B/8
def vit_forward(sd, rgb, prefix="", depth=12, num_heads=12, patch=8, eps=1e-6):
    """rgb [B,3,224,224] -> [B,768,28,28] (models/models.py:41-52)."""
    x = F.conv2d(rgb, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=patch)
    B, C, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + sd[prefix + "pos_embed"]
    for i in range(depth):
        x = _block(x, sd, f"{prefix}blocks.{i}", num_heads, eps)
    x = _ln(x, sd, prefix + "norm", eps)
    return x[:, 1:].permute(0, 2, 1).reshape(B, C, gh, gw)


# ----------------------------------------------------------------------------- Point-MAE
def _bn_eval(x, sd, prefix, batch_stats):
    if batch_stats:  # the reference as shipped never calls .eval()
        return F.batch_norm(x, None, None, sd[prefix + ".weight"], sd[prefix + ".bias"], True, 0.0, 1e-5)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, 1e-5)


def pointmae_encoder(sd, neighborhood, prefix="encoder.", batch_stats=False):
    """neighborhood [B,G,M,3] -> tokens [B,G,C] (models/models.py:200-215)."""
    B, G, M, _ = neighborhood.shape
    x = neighborhood.reshape(B * G, M, 3).transpose(2, 1)
    f = F.conv1d(x, sd[prefix + "first_conv.0.weight"], sd[prefix + "first_conv.0.bias"])
    f = F.relu(_bn_eval(f, sd, prefix + "first_conv.1", batch_stats))
    f = F.conv1d(f, sd[prefix + "first_conv.3.weight"], sd[prefix + "first_conv.3.bias"])
    g = f.max(dim=2, keepdim=True)[0]
    f = torch.cat([g.expand(-1, -1, M), f], dim=1)
    f = F.conv1d(f, sd[prefix + "second_conv.0.weight"], sd[prefix + "second_conv.0.bias"])
    f = F.relu(_bn_eval(f, sd, prefix + "second_conv.1", batch_stats))
    f = F.conv1d(f, sd[prefix + "second_conv.3.weight"], sd[prefix + "second_conv.3.bias"])
    return f.max(dim=2)[0].reshape(B, G, -1)


def pointmae_transformer(sd, tokens, center, prefix="", depth=12, num_heads=6, taps=(3, 11), eps=1e-5):
    """tokens [B,G,384], center [B,G,3] -> [B,768,G] (models/models.py:234-243, 360-373)."""
    pos = _linear(F.gelu(_linear(center, sd, prefix + "pos_embed.0")), sd, prefix + "pos_embed.2")
    x = tokens
    outs = []
    for i in range(depth):
        x = _block(x + pos, sd, f"{prefix}blocks.blocks.{i}", num_heads, eps)  # pos re-added every layer
        if i in taps:
            outs.append(_ln(x, sd, prefix + "norm", eps).transpose(-1, -2))
    return torch.cat(outs, dim=1)


def pointmae_forward(sd, neighborhood, center, prefix="", batch_stats=False):
    tok = pointmae_encoder(sd, neighborhood, prefix + "encoder.", batch_stats)
    return pointmae_transformer(sd, tok, center, prefix)


# ----------------------------------------------------------------------------- hallucination net
def halluc_generate(sd, x, direction):
    """direction 'xyz2rgb' = xyz_mlp(xyz_norm(x)); 'rgb2xyz' = rgb_mlp(rgb_norm(x)).
    models/hallucination_network.py:34-45, utils/utils.py:94-100 (GELU also on the output)."""
    name = "xyz" if direction == "xyz2rgb" else "rgb"
    h = _ln(x, sd, f"{name}_norm", 1e-5)
    d = 0
    while f"{name}_mlp.mlp_module.{d}.fc1.weight" in sd:      # mlp_depth chained blocks (utils/utils.py:110-113)
        p = f"{name}_mlp.mlp_module.{d}"
        h = F.gelu(_linear(h, sd, p + ".fc1"))
        h = F.gelu(_linear(h, sd, p + ".fc2"))
        h = F.gelu(_linear(h, sd, p + ".fc3"))
        d += 1
    return h


def halluc_losses(sd, xyz, rgb, dist_method="l2"):
    """(loss_xyz, loss_rgb) as hallucination_network.py:47-69."""
    xyz_h = halluc_generate(sd, rgb, "rgb2xyz")
    rgb_h = halluc_generate(sd, xyz, "xyz2rgb")
    B = xyz.shape[0]
    if dist_method == "l2":
        return (torch.linalg.norm(xyz_h - xyz, dim=2).sum() / B,
                torch.linalg.norm(rgb_h - rgb, dim=2).sum() / B)
    if dist_method == "cos_dist":
        return ((1 - F.cosine_similarity(xyz_h, xyz, dim=2)).sum() / B,
                (1 - F.cosine_similarity(rgb_h, rgb, dim=2)).sum() / B)
    if dist_method == "smooth_l1":
        return (F.smooth_l1_loss(xyz_h, xyz, reduction="none").sum() / B,
                F.smooth_l1_loss(rgb_h, rgb, reduction="none").sum() / B)
    raise NotImplementedError(dist_method)


# ----------------------------------------------------------------------------- synthetic weights
def _shapes_vit(prefix="", dim=768, depth=12, mlp=3072, tokens=785, patch=8):
    s = {prefix + "cls_token": (1, 1, dim), prefix + "pos_embed": (1, tokens, dim),
         prefix + "patch_embed.proj.weight": (dim, 3, patch, patch), prefix + "patch_embed.proj.bias": (dim,),
         prefix + "norm.weight": (dim,), prefix + "norm.bias": (dim,)}
    for i in range(depth):
        b = f"{prefix}blocks.{i}."
        s.update({b + "norm1.weight": (dim,), b + "norm1.bias": (dim,), b + "norm2.weight": (dim,),
                  b + "norm2.bias": (dim,), b + "attn.qkv.weight": (3 * dim, dim), b + "attn.qkv.bias": (3 * dim,),
                  b + "attn.proj.weight": (dim, dim), b + "attn.proj.bias": (dim,),
                  b + "mlp.fc1.weight": (mlp, dim), b + "mlp.fc1.bias": (mlp,),
                  b + "mlp.fc2.weight": (dim, mlp), b + "mlp.fc2.bias": (dim,)})
    return s


def _shapes_pointmae(prefix="", dim=384, depth=12):
    s = {prefix + "encoder.first_conv.0.weight": (128, 3, 1), prefix + "encoder.first_conv.0.bias": (128,),
         prefix + "encoder.first_conv.3.weight": (256, 128, 1), prefix + "encoder.first_conv.3.bias": (256,),
         prefix + "encoder.second_conv.0.weight": (512, 512, 1), prefix + "encoder.second_conv.0.bias": (512,),
         prefix + "encoder.second_conv.3.weight": (dim, 512, 1), prefix + "encoder.second_conv.3.bias": (dim,),
         prefix + "pos_embed.0.weight": (128, 3), prefix + "pos_embed.0.bias": (128,),
         prefix + "pos_embed.2.weight": (dim, 128), prefix + "pos_embed.2.bias": (dim,),
         prefix + "norm.weight": (dim,), prefix + "norm.bias": (dim,)}
    for name, c in (("first_conv.1", 128), ("second_conv.1", 512)):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{prefix}encoder.{name}.{leaf}"] = (c,)
    for i in range(depth):
        b = f"{prefix}blocks.blocks.{i}."
        s.update({b + "norm1.weight": (dim,), b + "norm1.bias": (dim,), b + "norm2.weight": (dim,),
                  b + "norm2.bias": (dim,), b + "attn.qkv.weight": (3 * dim, dim),
                  b + "attn.proj.weight": (dim, dim), b + "attn.proj.bias": (dim,),
                  b + "mlp.fc1.weight": (4 * dim, dim), b + "mlp.fc1.bias": (4 * dim,),
                  b + "mlp.fc2.weight": (dim, 4 * dim), b + "mlp.fc2.bias": (dim,)})
    return s


def _shapes_halluc(xyz_dim=768, rgb_dim=768, hidden_ratio=2.5, mlp_depth=1):
    s = {}
    for name, din, dout in (("xyz", xyz_dim, rgb_dim), ("rgb", rgb_dim, xyz_dim)):
        hid = int(din * hidden_ratio)
        s.update({f"{name}_norm.weight": (din,), f"{name}_norm.bias": (din,)})
        for d in range(mlp_depth):
            p = f"{name}_mlp.mlp_module.{d}."
            s.update({p + "fc1.weight": (hid, din), p + "fc1.bias": (hid,),
                      p + "fc2.weight": (hid, hid), p + "fc2.bias": (hid,),
                      p + "fc3.weight": (dout, hid), p + "fc3.bias": (dout,)})
    return s


def synth_state_dict(kind, seed, prefix="", **kw):
    """Deterministic, construction-order-independent synthetic weights: every tensor is drawn
    from its own generator seeded by (seed, crc32(name)).  Scales keep activations O(1) so the
    bf16 path is exercised at realistic magnitudes.  Shared by the golden script and the tests."""
    import zlib
    shapes = {"vit": _shapes_vit, "pointmae": _shapes_pointmae, "halluc": _shapes_halluc}[kind](
        **({"prefix": prefix} if kind != "halluc" else {}), **kw)
    sd = {}
    for name in sorted(shapes):
        shape = shapes[name]
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "weight" and len(shape) == 1:  # LayerNorm / BatchNorm gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            t = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("cls_token") or name.endswith("pos_embed"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[name] = t
    for name in list(sd):
        if name.endswith("running_var"):
            sd[name.replace("running_var", "num_batches_tracked")] = torch.tensor(0)
    return sd


def sharpen_pointmae(sd, conv_gain=400.0, qk_gain=36.0):
    """Synthetic Point-MAE weights whose features discriminate between patches (first-convolution gain, sharper attention
    logits): the generator lives beside the other synthetic inputs, cmdiad_amd/synth.py, shared by tests, goldens and bench."""
    from cmdiad_amd.synth import sharpen_pointmae as _sharpen
    return _sharpen(sd, conv_gain, qk_gain)


# ----------------------------------------------------------------------------- heavy-tailed synthetic weights
def outlier_vit(seed, channels=(7, 300, 555), gain=250.0, token=400, token_gain=400.0, from_block=2):
    """synth_state_dict("vit", seed) with the dynamic range of a real DINO checkpoint (models/models.py:35-53 loads
    `vit_base_patch8_224_dino`): from block `from_block` on a few residual CHANNELS carry values ~100x the typical one on every
    token ("massive activations": the block's fc2 bias feeds them, as in trained ViTs), and ONE token enters with ~8x the norm of
    the others (its positional embedding).  Downstream LayerNorms then see rows whose variance is three channels, attention sees one
    key far from the others, and the bf16 operand casts see 100x outliers beside O(1) values."""
    sd = synth_state_dict("vit", seed)
    b = sd[f"blocks.{from_block}.mlp.fc2.bias"].clone()
    for i, c in enumerate(channels):
        b[c] += gain * (1.0 if i % 2 == 0 else -0.8)
    sd[f"blocks.{from_block}.mlp.fc2.bias"] = b
    pe = sd["pos_embed"].clone()
    pe[0, token] *= token_gain
    sd["pos_embed"] = pe
    return sd


def outlier_pointmae(seed, channel=37, gain=50.0):
    """synth_state_dict("pointmae", seed) with ONE BatchNorm channel of the encoder's first point-MLP at 50x scale
    (models/models.py:188-190: first_conv.1): the second convolution's input then has one column 50x the others."""
    sd = synth_state_dict("pointmae", seed)
    w = sd["encoder.first_conv.1.weight"].clone()
    w[channel] *= gain
    sd["encoder.first_conv.1.weight"] = w
    return sd
