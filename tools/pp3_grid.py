#!/usr/bin/env python3
"""Persistent 256 x 256 network GEMM: blocks launched against time (test-only build, CMDIAD_PP3_GRID).  The vendor library runs these
shapes with grids that divide the tile count (198 blocks x 6 tiles for fc1) rather than one block per CU."""
import os as _os
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from tools.microbench import timeit
g = torch.Generator().manual_seed(0)
for name, M, N, K, act, grids in (("vit fc1 + GELU", 25120, 3072, 768, ops.ACT_GELU, (256, 238, 216, 198, 256, 238, 198)),
                                  ("vit qkv-like", 25120, 2304, 768, ops.ACT_NONE, (256, 223, 198, 179, 256, 223)),
                                  ("pmae fc1 + GELU", 32768, 1536, 384, ops.ACT_GELU, (256, 192, 154, 128, 256, 192))):
    A = torch.randn(M, K, generator=g).cuda().bfloat16(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda(); o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    os.environ["CMDIAD_GEMM_PP3"] = "1"
    tiles = ((M + 255) // 256) * (N // 256)
    for gr in grids:
        os.environ["CMDIAD_PP3_GRID"] = str(gr)
        ms = timeit(lambda: ops.gemm(A, W, bias=bias, act=act, out_bf16=o), iters=20, warm=3)
        abl = os.environ.get("CMDIAD_PP3_ABLATE", "0")
        print(f"ablate {abl} {name:16s} {tiles} tiles, grid {gr:3d} ({tiles / gr:.2f} tiles per block): {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:6.1f} TFLOP/s", flush=True)
