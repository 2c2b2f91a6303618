#!/usr/bin/env python3
"""Greedy coreset selection (features.py:372-425) at the bagel xyz size: 244 x 3136 = 765 184 rows, 334
projected dimensions (SparseRandomProjection eps = 0.9), fp16 rows -> n * d * 2 = 511 MB streamed per round.  Times
`rounds` rounds of cmdiad_coreset_greedy and reports ms per round and the scan rate against the 8 TB/s HBM peak
(511 MB exceed the 256 MB Infinity Cache: every round streams from HBM)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import coreset  # noqa: E402

n, d = int(os.environ.get("CS_N", 765184)), int(os.environ.get("CS_D", 334))
z = torch.randn(n, d, device="cuda")
for rounds in (51, 301):
    coreset.greedy_coreset(z, 11)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx = coreset.greedy_coreset(z, rounds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    coreset.greedy_coreset(z, 2)          # init pass (fp32 -> fp16 + first distances) + one round
    torch.cuda.synchronize()
    base = time.perf_counter() - t1
    per = (dt - base) / (rounds - 2)
    print(f"n={n} d={d}: {rounds} picks in {dt * 1e3:.1f} ms; {per * 1e6:.1f} us per round = {n * d * 2 / per / 1e12:.2f} TB/s "
          f"({n * d * 2 / per / 8e12:.2f} of the 8 TB/s HBM peak); full bagel selection (76 518 picks) ~ {per * 76517:.1f} s", flush=True)
assert len(set(idx.tolist())) == rounds
