#!/usr/bin/env python3
"""Timing of cmdiad_encoder_stage1 at the bench shape (32 x 1024 groups x 128 points)."""
import os as _os
# A/B tool: needs the test-only build with the superseded kernel formulations (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.runtime import fold_pointmae_encoder
from oracle import nets
from tools.microbench import timeit
w = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", "cuda")
groups, Mg = 32 * 1024, 128
nb = (0.02 * torch.randn(groups * Mg, 3, generator=torch.Generator().manual_seed(0))).cuda()
res = {}
for name, env in (("persistent", "1"), ("once-per-block", "0"), ("persistent", "1"), ("once-per-block", "0")):
    os.environ["CMDIAD_STAGE1_PERSIST"] = env
    out = ops.encoder_stage1(nb, w["w1b1"], w["W2"], w["b2"], groups, Mg)
    ms = timeit(lambda: ops.encoder_stage1(nb, w["w1b1"], w["W2"], w["b2"], groups, Mg), iters=10, warm=2)
    print(f"encoder_stage1 {name:15s} {ms:.3f} ms  h2 write {groups * Mg * 256 * 2 / ms / 1e6:.0f} GB/s", flush=True)
    res[name] = [o.clone() for o in out if torch.is_tensor(o)]
same = all(torch.equal(a, b) for a, b in zip(res["persistent"], res["once-per-block"]))
print("identical outputs (h2, group maxima):", same, flush=True)
for g2, mg in ((1024, 128), (100, 32), (7, 64), (256, 64)):   # small / ragged cases (M % 128 != 0 keeps the once kernel)
    nb2 = nb[: g2 * mg]
    os.environ["CMDIAD_STAGE1_PERSIST"] = "1"; a = ops.encoder_stage1(nb2, w["w1b1"], w["W2"], w["b2"], g2, mg)
    a = [o.clone() for o in a if torch.is_tensor(o)]
    os.environ["CMDIAD_STAGE1_PERSIST"] = "0"; b = ops.encoder_stage1(nb2, w["w1b1"], w["W2"], w["b2"], g2, mg)
    print(f"groups {g2} x {mg}: identical", all(torch.equal(x, y) for x, y in zip(a, b)), flush=True)
