import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit, line
g = torch.Generator().manual_seed(0); B = 32
for (T, H, nm) in [(785, 12, "vit"), (1024, 6, "pmae")]:
    Tp = (T + 63) // 64 * 64
    q = torch.randn(B, H, Tp, 64, generator=g).cuda().bfloat16(); k = torch.randn(B, H, Tp, 64, generator=g).cuda().bfloat16()
    vt = torch.randn(B, H, 64, Tp, generator=g).cuda().bfloat16()
    ms = timeit(lambda: ops.attention(q, k, vt, B, H, T), iters=20, warm=3)
    line(f"attention {nm} occ={os.environ.get('CMDIAD_ATT_OCC','2')}", ms, 4.0 * B * H * T * T * 64)
