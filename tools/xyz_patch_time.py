"""xyz_patch_fused (interpolated point features -> 56 x 56 patch rows) at the bench shape; CMDIAD_XYZ_PATCH_THREADS = threads per patch.
Prints the time and a checksum of the output bits (the variants must agree bit for bit)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops
from cmdiad_amd.synth import synth_cloud
from tools.microbench import timeit
B = 32
pcs = torch.cat([synth_cloud(1000 + i, 0.49) for i in range(B)]).cuda()          # 24 576 of 50 176 pixels are foreground, as in the bench
xyz, nz, pix2pt, nv = ops.unorganize(pcs, 34000)
idx, cen = ops.fps(xyz, 1024, nv)
idx3, w3 = ops.interp3nn(xyz, cen, nv)
feat = torch.randn(B, 1024, 768, generator=torch.Generator().manual_seed(1)).cuda()
out = ops.xyz_patch_fused(feat, idx3, w3, pix2pt, 224, 56, 0.1, 1.7)
out = out[0] if isinstance(out, tuple) else out
chk = int(out.view(torch.int32).to(torch.int64).sum().item())
for rep in range(3):
    ms = timeit(lambda: ops.xyz_patch_fused(feat, idx3, w3, pix2pt, 224, 56, 0.1, 1.7), iters=20, warm=3)
    print(f"threads/patch {os.environ.get('CMDIAD_XYZ_PATCH_THREADS', '256')}: {ms:.3f} ms  checksum {chk}", flush=True)
