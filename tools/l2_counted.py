#!/usr/bin/env python3
"""Distance GEMM with a device-resident live row count (cmdiad_l2_min_keys_counted) against the plain launch of the same rows:
what the blocks of the launched-but-dead query tiles cost."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit
Qmax, Nb, live = 100352, int(os.environ.get("L2_NB", 76544)), int(os.environ.get("L2_LIVE", 54401))
g = torch.Generator().manual_seed(0)
b16, _, bsq = ops.normalize_cast(torch.randn(Nb, 768, generator=g).cuda(), want_f32=False)
q16, _, qsq = ops.normalize_cast(torch.randn(Qmax, 768, generator=g).cuda(), want_f32=False)
keys = ops.new_keys(Qmax, "cuda")
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
for rnd in range(2):
    ms = timeit(lambda: ops.l2_min_keys(q16[:live], qsq[:live], b16, bsq, keys[:live]), iters=6, warm=2)
    print(f"plain   Q={live}: {ms:.3f} ms  {2.0 * live * Nb * 768 / ms / 1e9:.1f} TFLOP/s", flush=True)
    ms = timeit(lambda: ops.l2_min_keys_counted(q16, qsq, cnt, b16, bsq, keys), iters=6, warm=2)
    print(f"counted Q={live} of {Qmax}: {ms:.3f} ms  {2.0 * live * Nb * 768 / ms / 1e9:.1f} TFLOP/s", flush=True)
    ms = timeit(lambda: ops.l2_min_keys_counted(q16[:live], qsq[:live], cnt, b16, bsq, keys[:live]), iters=6, warm=2)
    print(f"counted Q={live} of {live}: {ms:.3f} ms  {2.0 * live * Nb * 768 / ms / 1e9:.1f} TFLOP/s", flush=True)
ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, keys), iters=6, warm=2)
print(f"plain   Q={Qmax}: {ms:.3f} ms  {2.0 * Qmax * Nb * 768 / ms / 1e9:.1f} TFLOP/s", flush=True)
