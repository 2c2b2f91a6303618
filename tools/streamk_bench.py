#!/usr/bin/env python3
"""Stream-K (csrc/gemm_sk.hip) against the 128 x 128 kernel on ViT-B/8's N = 768 residual products at batch 32:
    gpurun -- python tools/streamk_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops  # noqa: E402
from tools.microbench import line, timeit  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(0)
M = 32 * 785
for N, K, nm in ((768, 3072, "fc2"), (768, 768, "proj"), (768, 1536, "K=1536")):
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    x = torch.randn(M, N, generator=g).to(DEV)
    y = x.clone()
    ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=x, out_f32=x, want_bf16=False), iters=20, warm=3)
    line(f"128x128 residual {nm} {M}x{N}x{K}", ms, 2.0 * M * N * K)
    ms = timeit(lambda: ops.gemm_streamk(A, W, bias, y, out_f32=y), iters=20, warm=3)
    line(f"stream-K residual {nm} {M}x{N}x{K}", ms, 2.0 * M * N * K)
    os.environ["CMDIAD_GEMM_RES_WIDE"] = "1"      # the two-group kernel, one whole tile per block (read per call)
    ms = timeit(lambda: ops.gemm(A, W, bias=bias, residual=x, out_f32=x, want_bf16=False), iters=20, warm=3)
    line(f"256x256 tile per block residual {nm} {M}x{N}x{K}", ms, 2.0 * M * N * K)
    r = torch.randn(M, N, generator=g).to(DEV)
    a1, _ = ops.gemm(A, W, bias=bias, residual=r, want_f32=True, want_bf16=False)
    os.environ["CMDIAD_GEMM_RES_WIDE"] = "0"
    a0, _ = ops.gemm(A, W, bias=bias, residual=r, want_f32=True, want_bf16=False)
    print("   identical to the 128x128 kernel:", bool(torch.equal(a0, a1)), flush=True)
