#!/usr/bin/env python3
"""Kernel-only durations of the re-weighting scan (and the encoder's first stage) for a kernel trace:
rocprofv3 --kernel-trace --stats -- python3 tools/scan_profile.py"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.runtime import fold_pointmae_encoder  # noqa: E402
from oracle import nets  # noqa: E402

g = torch.Generator().manual_seed(0)
for rows in (76518, 19129):
    bank = torch.randn(rows, 768, generator=g).cuda()
    blk = ops.bank_block16(bank)
    probes = bank[:32].contiguous()
    for _ in range(25):
        ops.reweight_scan(probes, bank, blk)
    torch.cuda.synchronize()
w = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", "cuda")
nb = torch.randn(32 * 1024 * 128, 3, generator=g).cuda() * 0.01
for _ in range(10):
    ops.encoder_stage1(nb, w["w1b1"], w["W2"], w["b2"], 32 * 1024, 128)
torch.cuda.synchronize()
print("done")
