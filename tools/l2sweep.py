import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit, line
g = torch.Generator().manual_seed(0)
for Q, Nb in [(100352, 2048), (100352, 8192), (100352, 32768), (100352, 76518), (25088, 76518), (8192, 76518)]:
    bank = torch.randn(Nb, 768, generator=g).cuda(); qq = torch.randn(Q, 768, generator=g).cuda()
    b16, b32, bsq = ops.normalize_cast(bank, want_f32=True); q16, q32, qsq = ops.normalize_cast(qq, want_f32=True)
    keys = ops.new_keys(Q, "cuda")
    ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, keys), iters=5, warm=2)
    line(f"l2_min Q={Q} Nb={Nb}", ms, 2.0 * Q * Nb * 768)
