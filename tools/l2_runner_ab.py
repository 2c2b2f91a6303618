#!/usr/bin/env python3
"""Same-box A/B of the distance GEMM with and without the runner-up (round 6): the committed library against a side build of the
previous round's kernels (cmdiad_amd/libcmdiad_hip_r5.so, `git archive <round-5 commit> | make OUT=...`), both called through
ctypes with their own signatures, alternating, HIP events around every launch.
    python tools/l2_runner_ab.py [passes]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cmdiad_amd")
new = ctypes.CDLL(os.path.join(HERE, "libcmdiad_hip.so"))
old_path = os.path.join(HERE, "libcmdiad_hip_r5.so")
old = ctypes.CDLL(old_path) if os.path.exists(old_path) else None
P, I, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
new.cmdiad_l2_min_keys_counted.argtypes = [P, P, P, I, P, P, I, I, U, P, P, I, P]
if old is not None:
    old.cmdiad_l2_min_keys_counted.argtypes = [P, P, P, I, P, P, I, I, U, P, I, P]

Qmax, Nb, D = 100352, 76544, 768
g = torch.Generator().manual_seed(0)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for dt in (torch.bfloat16, torch.float16):
    b16, _, bsq = ops.normalize_cast(torch.randn(Nb, D, generator=g).cuda(), dtype=dt)
    q16, _, qsq = ops.normalize_cast(torch.randn(Qmax, D, generator=g).cuda(), dtype=dt)
    code = 1 if dt == torch.float16 else 0
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731

    def run(which, live, keys, keys2):
        cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
        keys.fill_(ops.KEY_EMPTY); keys2.fill_(ops.KEY_EMPTY)
        ts = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if which == "r5":
                rc = old.cmdiad_l2_min_keys_counted(p(q16), p(qsq), p(cnt), Qmax, p(b16), p(bsq), Nb, D, 0, p(keys), code, st)
            elif which == "r6 best only":
                rc = new.cmdiad_l2_min_keys_counted(p(q16), p(qsq), p(cnt), Qmax, p(b16), p(bsq), Nb, D, 0, p(keys), None, code, st)
            else:
                rc = new.cmdiad_l2_min_keys_counted(p(q16), p(qsq), p(cnt), Qmax, p(b16), p(bsq), Nb, D, 0, p(keys), p(keys2), code, st)
            assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts = sorted(ts[2:])
        return ts[len(ts) // 2]

    ka, kb, k2 = (torch.empty(Qmax, dtype=torch.int64, device="cuda") for _ in range(3))
    for live in (54401, Qmax):
        for ps in range(passes):
            row = []
            for which in (["r5"] if old is not None else []) + ["r6 best only", "r6 best + runner-up"]:
                ms = run(which, live, ka if which == "r5" else kb, k2)
                row.append(f"{which}: {ms:.3f} ms ({2.0 * live * Nb * D / ms / 1e9:.0f} TFLOP/s)")
            print(f"{str(dt).split('.')[-1]} live={live} pass {ps}: " + "   ".join(row), flush=True)
        if old is not None:
            assert torch.equal(ka[:live], kb[:live]), "the best plane must equal the previous round's keys"
            print("  best plane identical to the round-5 kernel's keys; runner-up present for",
                  int((k2[:live] != ops.KEY_EMPTY).sum()), "of", live, "rows", flush=True)
