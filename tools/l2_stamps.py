#!/usr/bin/env python3
"""In-kernel stamps of the production distance GEMM (l2_min_pp3_kernel, test-only build): where the cycles of a phase go.
Waves 0 (first group, bank-stream issuer) and 4 (second group, query-stream issuer) of one workgroup stamp s_memtime at five
points of every phase: phase start | fragment reads + LDS-DMA pieces issued | counted wait done | first barrier passed | MFMAs
issued (then the second barrier, = next phase start).  Printed: mean cycles per segment and phase kind over 60 K-tiles."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("CMDIAD_HIP_LIB", os.path.join(os.getcwd(), "cmdiad_amd", "libcmdiad_hip_ab.so"))
from cmdiad_amd import ops, _native

Q, Nb = int(os.environ.get("L2_Q", 100352)), 76518 // 256 * 256
g = torch.Generator().manual_seed(0)
bank = torch.randn(Nb, 768, generator=g).cuda(); qq = torch.randn(Q, 768, generator=g).cuda()
b16, _, bsq = ops.normalize_cast(bank, want_f32=True); q16, _, qsq = ops.normalize_cast(qq, want_f32=True)
keys = ops.new_keys(Q, "cuda")
lib = ctypes.CDLL(os.environ["CMDIAD_HIP_LIB"])
P = ctypes.c_void_p
lib.cmdiad_l2_diag.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P, P]
lib.cmdiad_l2_diag.restype = ctypes.c_int
N = 1280
names = ["reads+issue", "counted wait", "arrive->barrier 1", "MFMA issue", "barrier 2"]
for wg in [int(x) for x in os.environ.get("L2_WGS", "5000,9000").split(",")]:
    st = torch.zeros(2, N, dtype=torch.int32, device="cuda")
    for _ in range(3):   # warm clocks / caches, keep the last
        rc = lib.cmdiad_l2_diag(q16.data_ptr(), qsq.data_ptr(), b16.data_ptr(), bsq.data_ptr(), Q, Nb, 768, keys.data_ptr(), wg, st.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    for w, name in ((0, "wave 0 (first group, bank issuer)"), (1, "wave 4 (second group, query issuer)")):
        t = s[w]
        n = int((t != 0).sum()) // 20 * 20
        d = np.diff(t[:n + 1] if n < N else t[:n]) & 0xFFFFFFFF
        d = d[: (len(d) // 20) * 20].reshape(-1, 4, 5)[2:]       # [K-tile, phase, segment], first two K-tiles dropped
        print(f"workgroup {wg}, {name}: {d.shape[0]} K-tiles, {d.sum(axis=(1, 2)).mean():.0f} cycles per K-tile "
              f"(4 phases; 1024 = the two waves of a SIMD issuing MFMAs back to back)")
        for ph in range(4):
            print(f"  phase {ph}: " + "  ".join(f"{names[k]} {d[:, ph, k].mean():6.0f}" for k in range(5)) + f"   total {d[:, ph].sum(axis=1).mean():6.0f}")
