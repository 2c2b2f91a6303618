#!/usr/bin/env python3
"""Timings of the distillation-head kernels and of each head end to end, batch MB_BATCH (default 32).
`gpurun -- python tools/heads_bench.py`; one line per kernel / head with ms and the algorithmic rate."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.models import hallucination_network as hn  # noqa: E402
from cmdiad_amd.models.hrnet import HRNet  # noqa: E402
from tools.microbench import line, timeit  # noqa: E402

DEV = "cuda"


def main():
    B = int(os.environ.get("MB_BATCH", 32))
    g = torch.Generator().manual_seed(0)
    for (H, C, N, ks, st, nm) in [(56, 768, 768, 3, 1, "FtoF conv 3x3"), (56, 768, 384, 3, 1, "FtoI conv1"), (224, 384, 128, 3, 1, "FtoI conv2"),
                                  (224, 128, 64, 3, 1, "FtoI conv3"), (56, 128, 128, 3, 1, "bottleneck 3x3"),
                                  (56, 512, 128, 1, 1, "bottleneck 1x1 in"), (56, 128, 512, 1, 1, "bottleneck 1x1 out"),
                                  (112, 64, 128, 3, 2, "stem conv2 s2")]:
        b = B if H < 224 else min(B, 8)
        x = torch.randn(b, H, H, C, generator=g).to(DEV).bfloat16()
        w = (torch.randn(N, ks * ks * C, generator=g) / (ks * ks * C) ** 0.5).to(DEV).bfloat16()
        Ho = (H - 1) // st + 1
        out = torch.empty(b, Ho, Ho, N, device=DEV, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.conv2d_nhwc(x, w, N, ks, st, act=ops.ACT_RELU, out_bf16=out, want_bf16=False), iters=5)
        line(f"{nm} B={b} {H}x{H}x{C}->{N}", ms, 2.0 * b * Ho * Ho * N * ks * ks * C)
    x = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    w, bb = torch.randn(64, 3, 3, 3, generator=g).to(DEV), torch.randn(64, generator=g).to(DEV)
    ms = timeit(lambda: ops.conv_stem(x, w, bb, 2))
    line(f"stem conv1 B={B}", ms, 2.0 * B * 112 * 112 * 64 * 27, B * (3 * 224 * 224 * 4 + 112 * 112 * 64 * 2))
    b = min(B, 8)
    f = torch.randn(b, 56, 56, 384, generator=g).to(DEV)
    o = torch.empty(b, 224, 224, 384, device=DEV, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.upsample_bicubic(f, 384, 224, 224, out_bf16=o))
    line(f"bicubic 56->224 x384 B={b}", ms, None, b * (56 * 56 * 384 * 4 + 224 * 224 * 384 * 2))

    tok = torch.randn(B, 3136, 768, generator=g).to(DEV)
    img = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    m = hn.HallucinationCrossModalityConv(None, 768, 768).to(DEV).eval()
    ms = timeit(lambda: m.hallucination_generation(None, tok, "xyz"), iters=3)
    line(f"head conv FtoF (one direction) B={B}", ms, 4 * 2.0 * B * 3136 * 768 * 6912)
    m = HRNet(512, 768, 0.1).to(DEV).eval()
    ms = timeit(lambda: m.hallucination_tokens(img), iters=3)
    per_img = 2.0 * (112 * 112 * 64 * 27 + 3136 * 128 * 576 + 3136 * 512 * 128 + 12 * 3136 * (128 * 128 * 9 + 2 * 512 * 128) - 3136 * 384 * 128 + 3136 * 768 * 512)
    line(f"head HRNet ItoF B={B}", ms, B * per_img)
    b = min(B, 8)
    m = hn.HallucinationFeatureToInputConv(None, 768).to(DEV).eval()
    ms = timeit(lambda: m.hallucination_generation(tok[:b]), iters=3)
    line(f"head FtoI conv B={b}", ms, b * 2.0 * (3136 * 384 * 6912 + 50176 * (96 * 3456 + 32 * 864 + 3 * 288)))
    m = hn.HallucinationRGBFeatureToXYZInputMLP(types.SimpleNamespace(estimate_depth=False), 768).to(DEV).eval()
    ms = timeit(lambda: m.hallucination_generation(tok), iters=3)
    line(f"head FtoI MLP B={B}", ms, B * 3136 * 2.0 * (768 * 1152 + 1152 * 384 + 384 * 96 + 96 * 3))


if __name__ == "__main__":
    main()
