"""Attention kernel alone at the two production shapes.  `scale` multiplies q: 0.18 = head_dim^-0.5 * log2(e), what cmdiad_gemm_qkv
folds into q (scores ~ N(0, 1.4)); 1.0 = raw N(0, 64) scores (nearly one-hot softmax, maxima that keep growing: the rescale path)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit, line
g = torch.Generator().manual_seed(0); B = 32
for (T, H, nm) in [(785, 12, "vit"), (1024, 6, "pmae")]:
    Tp = (T + 63) // 64 * 64
    for scale in (0.18, 1.0):
        q = (torch.randn(B, H, Tp, 64, generator=g) * scale).cuda().bfloat16(); k = torch.randn(B, H, Tp, 64, generator=g).cuda().bfloat16()
        vt = torch.randn(B, H, 64, Tp, generator=g).cuda().bfloat16()
        ms = timeit(lambda: ops.attention(q, k, vt, B, H, T), iters=30, warm=5)
        line(f"attention {nm} q x {scale} lib={os.path.basename(os.environ.get('CMDIAD_HIP_LIB', 'production'))}", ms, 4.0 * B * H * T * T * 64)
