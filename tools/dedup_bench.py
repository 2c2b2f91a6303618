#!/usr/bin/env python3
"""Time of cmdiad_rows_dedup_plan + cmdiad_keys_expand at the bench's xyz query shape (100 352 x 768, 46 % repeats)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit
Q, D = 100352, 768
g = torch.Generator().manual_seed(0)
x = torch.randn(Q, D, generator=g)
x[torch.randperm(Q, generator=g)[:45951]] = -0.25
q16, _, qsq = ops.normalize_cast(x.cuda())
plan = ops.rows_dedup_plan(q16, qsq)
print("live rows", int(plan.count.item()))
ms = timeit(lambda: ops.rows_dedup_plan(q16, qsq, plan), iters=20, warm=3)
print(f"rows_dedup_plan: {ms * 1e3:.1f} us")
kc = ops.new_keys(Q, "cuda"); k = torch.empty_like(kc)
ms = timeit(lambda: ops.keys_expand(kc, plan.slot, k), iters=20, warm=3)
print(f"keys_expand: {ms * 1e3:.1f} us")
