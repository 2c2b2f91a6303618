#!/usr/bin/env python3
"""Same-process A/B of the two kNN-grouping formulations (CMDIAD_KNN_WAVE) at the bench shape; checks identical outputs."""
import os as _os
# A/B tool: needs the test-only build with the superseded kernel formulations (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.synth import synth_cloud_fixed_n
from tools.microbench import timeit
for B, N in ((32, 24576), (1, 24576), (8, 50176), (32, 8000)):
    pcs = [synth_cloud_fixed_n(100 + i, min(N, 50176)) for i in range(min(B, 4))]
    xyz = torch.stack([pc[0].reshape(3, -1).T[pc[0].reshape(3, -1).T.abs().sum(1) > 0][:N] for pc in pcs]).cuda().contiguous()
    xyz = xyz.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
    idx, cen = ops.fps(xyz, 1024)
    out = {}
    for v in ("0", "1"):
        os.environ["CMDIAD_KNN_WAVE"] = v
        ms = timeit(lambda: ops.knn_group(xyz, cen, 128), iters=5, warm=2)
        out[v] = (ms,) + tuple(ops.knn_group(xyz, cen, 128))
    same = torch.equal(out["0"][1], out["1"][1]) and torch.equal(out["0"][2], out["1"][2])
    print(f"B={B} N={xyz.shape[1]}: block {out['0'][0]:.3f} ms  wave {out['1'][0]:.3f} ms  identical: {same}", flush=True)
