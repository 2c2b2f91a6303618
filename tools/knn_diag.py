"""kNN grouping against the oracle on the parity tests' first case; CMDIAD_HIP_LIB selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from cmdiad_amd import ops
from oracle import kernels as ok
from test_gpu_kernels import _cloud
for frac, G, K in ((0.06, 64, 32), (0.3, 128, 128), (0.45, 1024, 128)):
    xyz, _ = _cloud(7, frac)
    _, cen = ok.fps(xyz[None], G)
    idx_ref, nb_ref = ok.knn_group(xyz[None], cen, K)
    idx, nb = ops.knn_group(torch.from_numpy(xyz[None]).cuda(), torch.from_numpy(cen).cuda(), K)
    bad = (idx.cpu().numpy() != idx_ref)
    print(os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1], f"n={len(xyz)} G={G} K={K}: {int(bad.sum())} of {bad.size} differ; centres with a difference: {int(bad.any(-1).sum())}", flush=True)
    if bad.any():
        g = int(np.argwhere(bad.any(-1))[0][1])
        missing = sorted(set(idx_ref[0, g].tolist()) - set(idx[0, g].cpu().tolist()))
        print("   first bad centre", g, "missing points", missing[:20], "steps", sorted(set(m // 128 for m in missing)))
