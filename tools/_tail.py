import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from cmdiad_amd import ops
from cmdiad_amd.runtime import fold_pointmae_encoder
from oracle import nets
from microbench import timeit
w = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", "cuda")
groups, Mg = 32 * 1024, 128
g = torch.Generator().manual_seed(0)
h2 = torch.randn(groups * Mg, 256, generator=g).cuda().bfloat16(); gb = torch.randn(groups, 512, generator=g).cuda()
def two():
    _, h3 = ops.gemm(h2, w["W3b"], act=ops.ACT_RELU, group_bias=gb, group_rows=Mg)
    return ops.gemm_groupmax(h3, w["W4"], w["b4"], groups, Mg)[0]
a = two(); b = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg)
print("equal", torch.equal(a, b))
print("two kernels ms", round(timeit(two, iters=5, warm=2), 3))
print("fused tail ms", round(timeit(lambda: ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg), iters=5, warm=2), 3))
