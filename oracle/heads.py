"""Torch-CPU fp32 restatement of the convolution / feature-to-input / HRNet distillation heads.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/ and tests/golden/make_golden.py, never by cmdiad_amd/.

Pure functions over a ``state_dict`` with the reference's parameter names, eval-mode semantics (BatchNorm running
statistics; the reference calls ``self.fusion.eval()`` after loading a checkpoint, features.py:107-111).  Pinned by
tests/golden/g10_heads.npz, produced by running the reference's own modules on the same synthetic weights.

Reference anchors:
  * conv_ftof ....... models/hallucination_network.py:72-131 (HallucinationCrossModalityConv)
  * ftoi_mlp ........ models/hallucination_network.py:146-172 (HallucinationRGBFeatureToXYZInputMLP)
  * ftoi_conv ....... models/hallucination_network.py:185-209 (HallucinationFeatureToInputConv)
  * hrnet ........... models/hrnet.py:8-43 (Bottleneck), 146-177 (constructed layers), 251-288 (the layers it runs)
"""
import math
import zlib

import torch
import torch.nn.functional as F


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _to_map(tokens):  # hallucination_network.py:6-9
    B, T, C = tokens.shape
    s = int(round(math.sqrt(T)))
    return tokens.transpose(1, 2).reshape(B, C, s, s)


def _to_tokens(fmap):  # hallucination_network.py:12-15
    return fmap.reshape(fmap.shape[0], fmap.shape[1], -1).transpose(1, 2)


def conv_ftof(sd, tokens, src):
    """src='rgb': rgb features -> hallucinated xyz features (rgb_conv); src='xyz': the reverse.  [B,3136,768] -> same."""
    x = _to_map(tokens)
    for i in range(4):
        x = F.conv2d(x, sd[f"{src}_conv.{3 * i}.weight"], None, stride=1, padding=1)
        if i < 3:
            x = F.relu(_bn(x, sd, f"{src}_conv.{3 * i + 1}"))
    return _to_tokens(x)


def ftoi_mlp(sd, tokens):
    """[B,3136,768] -> [B,out_dim,224,224]."""
    x = F.layer_norm(tokens, (tokens.shape[-1],), sd["rgb_norm.weight"], sd["rgb_norm.bias"], 1e-5)
    for i in (0, 2, 4, 6):
        x = F.linear(x, sd[f"mlp.{i}.weight"], sd[f"mlp.{i}.bias"])
        if i < 6:
            x = F.gelu(x)
    x = x.transpose(1, 2)
    x = x.reshape(x.shape[0], x.shape[1], 56, 56)
    return F.interpolate(x, size=(224, 224), mode="bicubic")


def ftoi_conv(sd, tokens):
    """[B,3136,768] -> [B,3,224,224] (the module's LayerNorm is never applied by the reference's forward)."""
    x = F.conv2d(_to_map(tokens), sd["conv1.weight"], sd["conv1.bias"], padding=1)
    x = F.interpolate(x, size=(224, 224), mode="bicubic")
    x = F.relu(F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], padding=1))
    x = F.relu(F.conv2d(x, sd["conv3.weight"], sd["conv3.bias"], padding=1))
    return F.conv2d(x, sd["conv4.weight"], sd["conv4.bias"], padding=1)


def _bottleneck(sd, x, p):
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2"))
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    if (p + "downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + "downsample.0.weight"]), sd, p + "downsample.1")
    return F.relu(out + x)


def hrnet(sd, img):
    """[B,3,224,224] -> [B,768,56,56] (layer4 is in the state_dict but not on the path, hrnet.py:260-262)."""
    x = F.relu(_bn(F.conv2d(img, sd["conv1.weight"], stride=2, padding=1), sd, "bn1"))
    x = F.relu(_bn(F.conv2d(x, sd["conv2.weight"], stride=2, padding=1), sd, "bn2"))
    for layer in (1, 2, 3):
        for i in range(4):
            x = _bottleneck(sd, x, f"layer{layer}.{i}.")
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"])


def mean_row_norm(a, b, dim):
    """The loss every head's forward returns: sum of L2 norms along `dim`, divided by the batch size."""
    d = torch.linalg.norm(a - b, dim=dim)
    return d.sum() / d.shape[0]


# ----------------------------------------------------------------------------- synthetic weights
def _bn_shapes(s, p, c):
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        s[f"{p}.{leaf}"] = (c,)


def head_shapes(kind, out_dim=3):
    s = {}
    if kind == "conv_ftof":
        for name in ("xyz", "rgb"):
            for i in range(4):
                s[f"{name}_conv.{3 * i}.weight"] = (768, 768, 3, 3)
                if i < 3:
                    _bn_shapes(s, f"{name}_conv.{3 * i + 1}", 768)
    elif kind == "ftoi_mlp":
        s.update({"rgb_norm.weight": (768,), "rgb_norm.bias": (768,)})
        for i, (a, b) in zip((0, 2, 4, 6), ((768, 1152), (1152, 384), (384, 96), (96, out_dim))):
            s[f"mlp.{i}.weight"], s[f"mlp.{i}.bias"] = (b, a), (b,)
    elif kind == "ftoi_conv":
        s.update({"norm.weight": (768,), "norm.bias": (768,)})
        for i, (a, b) in enumerate(((768, 384), (384, 96), (96, 32), (32, 3)), 1):
            s[f"conv{i}.weight"], s[f"conv{i}.bias"] = (b, a, 3, 3), (b,)
    elif kind == "hrnet":
        s["conv1.weight"], s["conv2.weight"] = (64, 3, 3, 3), (128, 64, 3, 3)
        _bn_shapes(s, "bn1", 64)
        _bn_shapes(s, "bn2", 128)
        for layer in (1, 2, 3, 4):
            for i in range(4):
                p, cin = f"layer{layer}.{i}.", (128 if (layer == 1 and i == 0) else 512)
                s[p + "conv1.weight"], s[p + "conv2.weight"], s[p + "conv3.weight"] = (128, cin, 1, 1), (128, 128, 3, 3), (512, 128, 1, 1)
                _bn_shapes(s, p + "bn1", 128)
                _bn_shapes(s, p + "bn2", 128)
                _bn_shapes(s, p + "bn3", 512)
                if layer == 1 and i == 0:
                    s[p + "downsample.0.weight"] = (512, 128, 1, 1)
                    _bn_shapes(s, p + "downsample.1", 512)
        s["final_layer.weight"], s["final_layer.bias"] = (768, 512, 1, 1), (768,)
    else:
        raise KeyError(kind)
    return s


def synth_head_state_dict(kind, seed, **kw):
    """Deterministic synthetic weights, one generator per tensor seeded by (seed, crc32(name)) as oracle.nets does.
    Convolution weights use gain sqrt(2) (He) so ReLU towers keep O(1) activations; the residual trunk's last BatchNorm
    of each Bottleneck is scaled down so twelve residual sums stay O(1)."""
    shapes = head_shapes(kind, **kw)
    sd = {}
    for name in sorted(shapes):
        shape = shapes[name]
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32((kind + "/" + name).encode())) % (2 ** 63))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "weight" and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            if name.endswith("bn3.weight"):
                t = 0.3 * t
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        sd[name] = t
    for name in list(sd):
        if name.endswith("running_var"):
            sd[name.replace("running_var", "num_batches_tracked")] = torch.tensor(0)
    return sd
