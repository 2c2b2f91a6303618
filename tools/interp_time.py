"""interp3nn at the bench shape (batch 32, 24 576 points, 1024 centres); CMDIAD_HIP_LIB selects the build to time."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.synth import synth_cloud_fixed_n
from tools.microbench import timeit
pcs = torch.cat([synth_cloud_fixed_n(1000 + i, 24576) for i in range(32)]).cuda()
xyz, nz, pix2pt, nv = ops.unorganize(pcs, 24576)
idx, cen = ops.fps(xyz, 1024, nv)
for rep in range(3):
    for grid in ("1", "0"):      # 1 = neighbourhood search on the binned centres (round 6), 0 = every centre for every point
        os.environ["CMDIAD_INTERP_GRID"] = grid
        ms = timeit(lambda: ops.interp3nn(xyz, cen, nv), iters=20, warm=3)
        print(os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1], f"interp3nn B=32 N=24576 S=1024 grid={grid}: {ms:.3f} ms", flush=True)
os.environ["CMDIAD_INTERP_GRID"] = "1"; a = ops.interp3nn(xyz, cen, nv)
os.environ["CMDIAD_INTERP_GRID"] = "0"; b = ops.interp3nn(xyz, cen, nv)
print("identical:", bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])), flush=True)
