#!/usr/bin/env python3
"""Same-process A/B of the network GEMMs: 128 x 128 two-blocks-per-CU kernel (CMDIAD_GEMM_PP3=0) against the two-group
persistent 256 x 256 kernel (=1, where legal: bias + bf16 output) on the transformer shapes at batch 32, WITH the epilogues
the networks use.  Checks identical outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(0)
shapes = [("vit qkv-like (bias, bf16 out)", 25120, 2304, 768, "bias"), ("vit fc1 (bias+GELU)", 25120, 3072, 768, "gelu"),
          ("vit fc2 (bias+residual f32)", 25120, 768, 3072, "res"), ("vit proj (bias+residual f32)", 25120, 768, 768, "res"),
          ("pmae fc1 (bias+GELU)", 32768, 1536, 384, "gelu"), ("pmae fc2 (residual)", 32768, 384, 1536, "res"),
          ("halluc fc2 (GELU) 100352x1920x1920", 100352, 1920, 1920, "gelu")]
for name, M, N, K, epi in shapes:
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    x = torch.randn(M, N, generator=g).to(DEV)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)

    def call():
        if epi == "res":
            ops.gemm(A, W, bias=bias, residual=x, out_f32=x, want_bf16=False)
        else:
            ops.gemm(A, W, bias=bias, act=ops.ACT_GELU if epi == "gelu" else ops.ACT_NONE, out_bf16=o16)
    res = {}
    outs = {}
    legal = N % 256 == 0
    modes = {"128x128": "0", "pp3": "1"} if legal else {"128x128": "0"}
    x0 = x.clone()
    for rnd in range(3):
        for mode, p3 in modes.items():
            os.environ["CMDIAD_GEMM_PP3"] = p3
            ms = timeit(call, iters=10, warm=2)
            res.setdefault(mode, []).append(ms)
            if rnd == 0:
                if epi == "res":
                    x.copy_(x0); call(); outs[mode] = x.clone(); x.copy_(x0)
                else:
                    outs[mode] = o16.clone()
    os.environ.pop("CMDIAD_GEMM_PP3", None)
    same = all(torch.equal(outs["128x128"], o) for o in outs.values())
    line = f"{name:42s}"
    for mode, v in res.items():
        v = sorted(v)[len(v) // 2]
        line += f"  {mode}: {v:7.3f} ms {2.0 * M * N * K / v / 1e9:7.1f} TF"
    print(line, " identical" if same else "  OUTPUTS DIFFER", flush=True)
