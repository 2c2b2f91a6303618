"""kNN grouping against the oracle at a grid that selects the production instantiation (knn_wave_kernel<4, 4>: B * ceil(G / 16) >= 512),
ragged clouds, repeated for determinism; CMDIAD_HIP_LIB selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from cmdiad_amd import ops
from oracle import kernels as ok
from test_gpu_kernels import _cloud
tag = os.environ.get("CMDIAD_HIP_LIB", "default").split("/")[-1]
B, G, K = 8, 1024, 128
clouds = [_cloud(20 + i, 0.05 + 0.01 * i)[0] for i in range(B)]
N = max(len(c) for c in clouds)
xyz = np.zeros((B, N, 3), np.float32)
for i, c in enumerate(clouds):
    xyz[i, :len(c)] = c
nv = torch.tensor([len(c) for c in clouds], dtype=torch.int32, device="cuda")
cen = np.stack([ok.fps(c[None], G)[1][0] for c in clouds])
ref = np.stack([ok.knn_group(c[None], cen[i:i + 1], K)[0][0] for i, c in enumerate(clouds)])
x, c = torch.from_numpy(xyz).cuda(), torch.from_numpy(cen).cuda()
first = None
for rep in range(6):
    idx, nb = ops.knn_group(x, c, K, n_valid=nv)
    got = idx.cpu().numpy()
    bad = got != ref
    same = True if first is None else bool(np.array_equal(got, first))
    first = got if first is None else first
    print(tag, f"rep {rep}: {int(bad.sum())} of {bad.size} indices differ from the oracle ({int(bad.any(-1).sum())} centres); same as rep 0: {same}", flush=True)
    if rep == 0 and bad.any():
        bc = np.argwhere(bad.any(-1))
        print("   bad centres by slot g % 4:", np.bincount(bc[:, 1] % 4, minlength=4).tolist(), " by cloud:", np.bincount(bc[:, 0], minlength=B).tolist())
        miss_h, miss_lane = np.zeros(2, int), np.zeros(64, int)
        for b_, g_ in bc[:200]:
            missing = sorted(set(ref[b_, g_].tolist()) - set(got[b_, g_].tolist()))
            for m_ in missing:
                miss_h[(m_ % 128) // 64] += 1
                miss_lane[m_ % 64] += 1
        print("   missing points by half step:", miss_h.tolist(), " by lane:", miss_lane.tolist())
        b_, g_ = bc[0]
        missing = sorted(set(ref[b_, g_].tolist()) - set(got[b_, g_].tolist()))
        extra = sorted(set(got[b_, g_].tolist()) - set(ref[b_, g_].tolist()))
        print(f"   cloud {b_} (n = {len(clouds[b_])}) centre {g_}: missing {missing[:24]}  extra {extra[:24]}")
        d = ((clouds[b_] - cen[b_, g_]) ** 2).sum(-1)
        print("   d2 of missing", np.round(d[missing[:8]], 6).tolist(), "d2 of extra", np.round(d[[e for e in extra[:8] if e < len(d)]], 6).tolist(), "K-th ref d2", float(np.sort(d)[K - 1]))
