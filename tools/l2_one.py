"""One configuration of the distance GEMM (cmdiad_l2_min_keys) at the bench's xyz-library shape, for PMC / clock passes:
CMDIAD_L2_QGROUP / CMDIAD_L2_SPLITS pick the block -> (query tile, library range) mapping (tools/l2_fetch_sweep.sh)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from microbench import timeit

Q = int(os.environ.get("L2_Q", 100352)); Nb = int(os.environ.get("L2_NB", 76518)); it = int(os.environ.get("L2_ITERS", 6))
g = torch.Generator().manual_seed(0)
bank = torch.randn(Nb, 768, generator=g).cuda(); qq = torch.randn(Q, 768, generator=g).cuda()
if os.environ.get("L2_ZEROS"): bank.zero_(); qq.zero_()
b16, _, bsq = ops.normalize_cast(bank, want_f32=True); q16, _, qsq = ops.normalize_cast(qq, want_f32=True)
keys = ops.new_keys(Q, "cuda")
ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, keys), iters=it, warm=2)
print(f"l2_one qgroup={os.environ.get('CMDIAD_L2_QGROUP','-')} splits={os.environ.get('CMDIAD_L2_SPLITS','-')} zeros={bool(os.environ.get('L2_ZEROS'))} "
      f"Q={Q} Nb={Nb}: {ms:.3f} ms  {2.0 * Q * Nb * 768 / ms / 1e9:.1f} TFLOP/s", flush=True)
