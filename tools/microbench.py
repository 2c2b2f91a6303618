#!/usr/bin/env python3
"""Per-kernel timings at the bench shapes (batch 32, bagel-sized banks).  Development aid:
`gpurun -- python tools/microbench.py`; prints one line per kernel with ms and the derived rate."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import ops, runtime  # noqa: E402
from cmdiad_amd.synth import synth_cloud_fixed_n  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def line(name, ms, flops=None, bytes_=None):
    s = f"{name:34s} {ms:9.3f} ms"
    if flops:
        s += f"  {flops / ms / 1e9:9.1f} TFLOP/s"
    if bytes_:
        s += f"  {bytes_ / ms / 1e6:9.1f} GB/s"
    print(s, flush=True)


def main():
    B = int(os.environ.get("MB_BATCH", 32))
    g = torch.Generator().manual_seed(0)
    # ---- GEMMs at ViT shapes
    M = B * 785
    for (N, K, nm) in [(2304, 768, "qkv"), (768, 768, "proj"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
        A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        W = torch.randn(N, K, generator=g).to(DEV).bfloat16()
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ms = timeit(lambda: ops.gemm(A, W, out_bf16=out))
        line(f"gemm vit {nm} {M}x{N}x{K}", ms, 2.0 * M * N * K)
    if os.environ.get("MB_ONLY") == "gemm":
        for (Qi, Nb, nm) in [(3136, 76518, "xyz")]:
            Q = B * Qi
            bank = torch.randn(Nb, 768, generator=g).to(DEV)
            qq = torch.randn(Q, 768, generator=g).to(DEV)
            b16, b32, bsq = ops.normalize_cast(bank, want_f32=True)
            q16, q32, qsq = ops.normalize_cast(qq, want_f32=True)
            keys = ops.new_keys(Q, DEV)
            ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, keys), iters=3, warm=1)
            line(f"l2_min_keys {nm} Q={Q} Nb={Nb}", ms, 2.0 * Q * Nb * 768)
        M2 = B * 1024 * 128
        for (N, K, nm) in [(512, 256, "enc h3"), (384, 512, "enc out")]:
            A = torch.randn(M2, K, generator=g).to(DEV).bfloat16()
            W = torch.randn(N, K, generator=g).to(DEV).bfloat16()
            out = torch.empty(M2, N, device=DEV, dtype=torch.bfloat16)
            ms = timeit(lambda: ops.gemm(A, W, out_bf16=out), iters=3, warm=1)
            line(f"gemm {nm} {M2}x{N}x{K}", ms, 2.0 * M2 * N * K)
        return
    # ---- attention
    for (T, H, nm) in [(785, 12, "vit"), (1024, 6, "pmae")]:
        Tp = (T + 63) // 64 * 64
        q = torch.randn(B, H, Tp, 64, generator=g).to(DEV).bfloat16()
        k = torch.randn(B, H, Tp, 64, generator=g).to(DEV).bfloat16()
        vt = torch.randn(B, H, 64, Tp, generator=g).to(DEV).bfloat16()
        ms = timeit(lambda: ops.attention(q, k, vt, B, H, T))
        line(f"attention {nm} T={T} H={H}", ms, 4.0 * B * H * T * T * 64)
    # ---- distance GEMM (xyz bank, bagel)
    for (Qi, Nb, nm) in [(3136, 76518, "xyz"), (784, 19129, "rgb")]:
        Q = B * Qi
        bank = torch.randn(Nb, 768, generator=g).to(DEV)
        qq = torch.randn(Q, 768, generator=g).to(DEV)
        b16, b32, bsq = ops.normalize_cast(bank, want_f32=True)
        q16, q32, qsq = ops.normalize_cast(qq, want_f32=True)
        keys = ops.new_keys(Q, DEV)
        ms = timeit(lambda: ops.l2_min_keys(q16, qsq, b16, bsq, keys), iters=3, warm=1)
        line(f"l2_min_keys {nm} Q={Q} Nb={Nb}", ms, 2.0 * Q * Nb * 768)
        ms = timeit(lambda: ops.l2_rescore(q32, b32, keys))
        line(f"l2_rescore {nm}", ms, None, Q * 768 * 8.0)
        probes = bank[:B].contiguous()
        blk = ops.bank_block16(b32)
        ms = timeit(lambda: ops.reweight_scan(probes, b32, blk), iters=10, warm=2)
        line(f"reweight_scan {nm} R={B}", ms, None, B * Nb * 768 * 4.0)
    # ---- point-cloud front end
    pcs = torch.cat([synth_cloud_fixed_n(100 + i, 24576) for i in range(B)], 0).to(DEV)
    ms = timeit(lambda: ops.unorganize(pcs, 24576))
    line("unorganize", ms)
    xyz, nz, pix2pt, nv = ops.unorganize(pcs, 24576)
    ms = timeit(lambda: ops.fps(xyz, 1024, nv), iters=3, warm=1)
    line(f"fps B={B} N=24576 G=1024", ms)
    idx, cen = ops.fps(xyz, 1024, nv)
    ms = timeit(lambda: ops.knn_group(xyz, cen, 128, nv), iters=3, warm=1)
    line("knn_group K=128", ms)
    ms = timeit(lambda: ops.interp3nn(xyz, cen, nv))
    line("interp3nn", ms)
    idx3, w3 = ops.interp3nn(xyz, cen, nv)
    feat = torch.randn(B, 1024, 768, generator=g).to(DEV)
    ms = timeit(lambda: ops.xyz_patch_fused(feat, idx3, w3, pix2pt, 224, 56, want_bf16=True))
    line("xyz_patch_fused P=56", ms, None, B * (3136 * 768 * 6.0 + 1024 * 768 * 4))
    # ---- whole networks
    from oracle import nets
    vit = runtime.PackedViT(nets.synth_state_dict("vit", 31), device=DEV)
    rgb = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    ms = timeit(lambda: vit.forward_tokens(rgb), iters=3, warm=1)
    line(f"ViT-B/8 forward B={B}", ms, B * 156.3e9)
    pm = runtime.PackedPointMAE(nets.synth_state_dict("pointmae", 21), device=DEV)
    _, nb = ops.knn_group(xyz, cen, 128, nv)
    ms = timeit(lambda: pm.encode(nb), iters=3, warm=1)
    line("Point-MAE encoder", ms, B * 1024 * 128 * 2.0 * (128 * 256 + 256 * 512 + 512 * 384))
    tok = pm.encode(nb)
    ms = timeit(lambda: pm.transform(tok.clone(), cen), iters=3, warm=1)
    line("Point-MAE transformer", ms, B * 62.8e9)
    ms = timeit(lambda: pm.forward(xyz, nv), iters=2, warm=1)
    line("Point-MAE full (fps+knn+enc+tr)", ms)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"total {time.time() - t0:.1f}s")
