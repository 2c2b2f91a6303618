#!/usr/bin/env python3
"""Timing of cmdiad_encoder_tail at the bench shape (32 x 1024 groups x 128 points) + equality with the two-kernel path."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.runtime import fold_pointmae_encoder
from oracle import nets
from tools.microbench import timeit
sd = nets.synth_state_dict("pointmae", 21)
w = fold_pointmae_encoder(sd, "encoder.", "cuda")
groups, Mg = 32 * 1024, 128
g = torch.Generator().manual_seed(0)
h2 = (torch.randn(groups * Mg, 256, generator=g) * 0.5).cuda().bfloat16()
gb = torch.randn(groups, 512, generator=g).cuda()
ms = timeit(lambda: ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg), iters=5, warm=2)
tok = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg)
_, h3 = ops.gemm(h2, w["W3b"], act=ops.ACT_RELU, group_bias=gb, group_rows=Mg)
ref, _ = ops.gemm_groupmax(h3, w["W4"], w["b4"], groups, Mg)
print(f"encoder_tail {ms:.3f} ms  {2.0 * groups * Mg * (256 * 512 + 512 * 384) / ms / 1e9:.0f} TFLOP/s  identical to the two-kernel path: {torch.equal(tok, ref)}", flush=True)
