#!/usr/bin/env python3
"""The 128 x 128 network GEMM on ViT-B/8 / Point-MAE shapes; CMDIAD_HIP_LIB selects the build (production, or a timing-only build
whose operand loads read hot sources: -DCMDIAD_ABL_HOT=1 every block the same tile, =2 a block's first K-tile again and again):
    gpurun -- 'for l in "" _hot1 _hot2; do CMDIAD_HIP_LIB=$PWD/cmdiad_amd/libcmdiad_hip$l.so python tools/gemm_hot_ab.py; done'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops  # noqa: E402
from tools.microbench import line, timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(0)
tag = os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1]
for M, N, K, nm in ((32 * 785, 768, 768, "vit proj"), (32 * 785, 768, 3072, "vit fc2"), (32 * 785, 2304, 768, "vit qkv-like"),
                    (32 * 785, 3072, 768, "vit fc1-like"), (32768, 384, 384, "pmae proj"), (32768, 1536, 384, "pmae fc1-like")):
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    ms = timeit(lambda: ops.gemm(A, W, bias=bias), iters=20, warm=3)
    line(f"[{tag}] {nm} {M}x{N}x{K} bf16 out", ms, 2.0 * M * N * K)
