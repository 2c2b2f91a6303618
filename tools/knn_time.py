"""kNN grouping at the bench shape (batch 32, 24 576 points, 1024 centres x 128): CMDIAD_HIP_LIB selects the build to time."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.synth import synth_cloud_fixed_n
from tools.microbench import timeit
pcs = torch.cat([synth_cloud_fixed_n(1000 + i, 24576) for i in range(32)]).cuda()
xyz, nz, pix2pt, nv = ops.unorganize(pcs, 24576)
idx, cen = ops.fps(xyz, 1024, nv)
ms = timeit(lambda: ops.fps(xyz, 1024, nv), iters=10, warm=3)
print(os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1], f"fps B=32 N=24576 G=1024: {ms:.3f} ms", flush=True)
for rep in range(3):
    for grid in ("1", "0"):      # 1 = neighbourhood search on the binned cloud (round 6), 0 = the streaming kernel
        os.environ["CMDIAD_KNN_GRID"] = grid
        ms = timeit(lambda: ops.knn_group(xyz, cen, 128, nv), iters=10, warm=3)
        print(os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1], f"knn B=32 N=24576 grid={grid}: {ms:.3f} ms", flush=True)
os.environ["CMDIAD_KNN_GRID"] = "1"; a = ops.knn_group(xyz, cen, 128, nv)
os.environ["CMDIAD_KNN_GRID"] = "0"; b = ops.knn_group(xyz, cen, 128, nv)
print("identical:", bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])), flush=True)
# ragged batch (the var_n leg's shape)
import numpy as np
from cmdiad_amd.synth import synth_cloud
rs = np.random.RandomState(8)
fr = (0.35 + 0.30 * rs.rand(32)) / 0.85
xyz_v, _, _, nv_v = ops.unorganize(torch.cat([synth_cloud(7000 + i, float(fr[i])) for i in range(32)]).cuda(), None)
_, cen_v = ops.fps(xyz_v, 1024, nv_v)
for grid in ("1", "0"):
    os.environ["CMDIAD_KNN_GRID"] = grid
    ms = timeit(lambda: ops.knn_group(xyz_v, cen_v, 128, nv_v), iters=10, warm=3)
    print(f"knn ragged ({int(nv_v.min())}..{int(nv_v.max())} points, padded {xyz_v.shape[1]}) grid={grid}: {ms:.3f} ms", flush=True)
