import os as _os
# A/B tool: needs the test-only build with the superseded kernel formulations (make -C cmdiad_amd/csrc ab)
_os.environ.setdefault("CMDIAD_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cmdiad_amd", "libcmdiad_hip_ab.so"))
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.synth import synth_cloud_fixed_n
from tools.microbench import timeit
for B, N in ((32, 24576), (1, 24576), (1, 16000), (1, 28000), (4, 8000)):
    xyz = torch.stack([synth_cloud_fixed_n(100 + i, N)[0].reshape(3, -1).T[synth_cloud_fixed_n(100 + i, N)[0].reshape(3, -1).T.abs().sum(1) > 0][:N] for i in range(min(B, 4))]).cuda().contiguous()
    xyz = xyz.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
    out = {}
    for v in ("0", "1"):
        os.environ["CMDIAD_FPS_PK"] = v
        ms = timeit(lambda: ops.fps(xyz, 1024), iters=5, warm=2)
        out[v] = (ms, ops.fps(xyz, 1024)[0])
    assert torch.equal(out["0"][1], out["1"][1])
    print(f"B={B} N={xyz.shape[1]}: reg {out['0'][0]:.3f} ms  pk {out['1'][0]:.3f} ms", flush=True)
