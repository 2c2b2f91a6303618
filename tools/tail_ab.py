#!/usr/bin/env python3
"""Timing of cmdiad_encoder_tail at the bench shape (32 x 1024 groups x 128 points) + equality with the two-kernel path."""
import os as _os
# A/B tool: needs the test-only build (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
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
_, h3 = ops.gemm(h2, w["W3b"], act=ops.ACT_RELU, group_bias=gb, group_rows=Mg)
ref, _ = ops.gemm_groupmax(h3, w["W4"], w["b4"], groups, Mg)
del h3
for name, env in (("persistent", ""), ("two-group", "1"), ("lock-step", "0"), ("persistent", ""), ("two-group", "1"), ("lock-step", "0")):
    os.environ["CMDIAD_TAIL_PP"] = env
    ms = timeit(lambda: ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg), iters=8, warm=2)
    tok = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg)
    print(f"encoder_tail {name:10s} {ms:.3f} ms  {2.0 * groups * Mg * (256 * 512 + 512 * 384) / ms / 1e9:.0f} TFLOP/s  identical to the two-kernel path: {torch.equal(tok, ref)}", flush=True)
if os.environ.get("TAIL_STRESS"):   # race screen: the same launch many times, every result compared
    os.environ["CMDIAD_TAIL_PP"] = ""
    bad = 0
    for it in range(int(os.environ["TAIL_STRESS"])):
        tok = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg)
        bad += int(not torch.equal(tok, ref))
    print("stress runs differing from the reference:", bad, flush=True)
if os.environ.get("TAIL_ABLATE"):   # timing only (results are garbage): which part of a phase the time is in
    os.environ["CMDIAD_TAIL_PP"] = ""
    names = {1: "no weight stream", 2: "no MFMAs", 4: "no fragment reads", 8: "no h3 write", 3: "no stream, no MFMAs", 6: "no MFMAs, no reads",
             5: "no stream, no reads", 7: "barriers only (+h3 write)", 15: "barriers only", 31: "prologue + final reduction only", 47: "barriers only, no h2 load", 63: "final reduction only", 16: "no phase barriers", 127: "launch + accumulators + store only", 79: "barriers only, no final reduction", 95: "loop skeleton only"}
    for bits in (0, 1, 2, 3, 15, 31, 79, 95):
        os.environ["CMDIAD_TAIL_ABLATE"] = str(bits)
        ms = timeit(lambda: ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg), iters=8, warm=2)
        print(f"ablate {bits:2d} ({names.get(bits, 'full')}): {ms:.3f} ms", flush=True)
    os.environ.pop("CMDIAD_TAIL_ABLATE")
