#!/usr/bin/env python3
"""Same-process A/B of the distance-GEMM tile variants (CMDIAD_L2_TILE) on the bench shape: interleaved rounds, median and min."""
import os as _os
# A/B tool: needs the test-only build with the superseded kernel formulations (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from tools.microbench import timeit
g = torch.Generator().manual_seed(0)
Q, Nb = int(os.environ.get("L2_Q", 100352)), int(os.environ.get("L2_NB", 76518))
bank = torch.randn(Nb, 768, generator=g).cuda(); qq = torch.randn(Q, 768, generator=g).cuda()
b16, b32, bsq = ops.normalize_cast(bank, want_f32=True); q16, q32, qsq = ops.normalize_cast(qq, want_f32=True)
res, keys = {}, {}
variants = os.environ.get("L2_VARIANTS", "2,5").split(",")
for rnd in range(5):
    for v in variants:
        os.environ["CMDIAD_L2_TILE"] = v
        k = ops.new_keys(Q, "cuda")
        ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, k), iters=3, warm=1)
        res.setdefault(v, []).append(ms)
        keys[v] = k
for v in variants:
    r = res[v]
    print(f"tile {v}: median {statistics.median(r):.3f} ms  min {min(r):.3f} ms  -> {2.0 * Q * Nb * 768 / statistics.median(r) / 1e9:.0f} TFLOP/s", flush=True)
a, b = keys[variants[0]], keys[variants[-1]]
print("argmin agreement between variants:", (a & 0xFFFFFFFF == b & 0xFFFFFFFF).float().mean().item(), " identical keys:", (a == b).float().mean().item())
n_stress = int(os.environ.get("L2_STRESS", "0"))
if n_stress:
    os.environ["CMDIAD_L2_TILE"] = variants[0]
    ref = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, "cuda")).clone()
    os.environ["CMDIAD_L2_TILE"] = variants[-1]
    bad = 0
    for i in range(n_stress):
        k = ops.l2_min_keys(q16, qsq, b16, bsq, ops.new_keys(Q, "cuda"))
        bad += int(not torch.equal(k, ref))
    print(f"stress: {n_stress} launches of tile {variants[-1]} against tile {variants[0]}: {bad} mismatching launches", flush=True)
