import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.runtime import fold_pointmae_encoder
from oracle import nets
w = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", "cuda")
for Mg, groups in ((128, 24), (128, 2), (64, 9)):
    g = torch.Generator().manual_seed(Mg + groups)
    h2 = torch.randn(groups * Mg, 256, generator=g).cuda().bfloat16()
    gb = torch.randn(groups, 512, generator=g).cuda()
    _, h3 = ops.gemm(h2, w["W3b"], act=ops.ACT_RELU, group_bias=gb, group_rows=Mg)
    want, _ = ops.gemm_groupmax(h3, w["W4"], w["b4"], groups, Mg)
    got = ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, Mg)
    torch.cuda.synchronize()
    d = (got - want)
    bad = (d != 0)
    print(Mg, groups, "bad", int(bad.sum()), "of", bad.numel(), "rows with bad", bad.any(1).nonzero().flatten().tolist()[:10],
          "cols", bad.any(0).nonzero().flatten().tolist()[:12], "maxdiff", float(d.abs().max()))
    if bad.any():
        i = bad.nonzero()[0]
        print(" first", i.tolist(), float(got[i[0], i[1]]), float(want[i[0], i[1]]), "b4", float(w["b4"][i[1]]), "diff-b4?", float(got[i[0], i[1]] - want[i[0], i[1]]))
