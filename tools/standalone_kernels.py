#!/usr/bin/env python3
"""Every kernel of DESIGN.md section 4 that the pipelined bench trace cannot time on its own, launched ALONE at the bench shapes
(batch 32, bagel-sized libraries), for a kernel trace and for counter passes:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4/standalone -- python3 tools/standalone_kernels.py
    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/prof_r4/standalone_pmc/fetch -- python3 tools/standalone_kernels.py hbm
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/prof_r4/standalone_pmc/write -- python3 tools/standalone_kernels.py hbm
    python tools/standalone_summary.py gpurun_out/prof_r4/standalone gpurun_out/prof_r4/standalone_pmc > profiles/r4_standalone.md

    rocprofv3 --pmc FETCH_SIZE  --output-format csv -d gpurun_out/prof_r5/standalone_pmc/l2fetch -- python3 tools/standalone_kernels.py l2
    rocprofv3 --pmc WRITE_SIZE  --output-format csv -d gpurun_out/prof_r5/standalone_pmc/l2write -- python3 tools/standalone_kernels.py l2

`hbm`: only the two HBM-bound kernels whose bytes are the claim (re-weighting scan, coreset round).  `l2`: only the distance GEMM
(the dominant kernel of bench.py's `roofline`) at its four shapes.  The algorithmic work of every launch is written to
standalone_work.json next to the trace (kernel-name prefix -> bytes / flops per launch; `seq` = (first, count) in the kernel's
launch order where one kernel name runs several shapes on the same grid)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import coreset, ops  # noqa: E402
from cmdiad_amd.runtime import fold_pointmae_encoder  # noqa: E402
from cmdiad_amd.synth import synth_cloud, synth_cloud_fixed_n  # noqa: E402

DEV = "cuda"
only_hbm = len(sys.argv) > 1 and sys.argv[1] == "hbm"
only_l2 = len(sys.argv) > 1 and sys.argv[1] == "l2"
B, G, K, D = 32, 1024, 128, 768
g = torch.Generator().manual_seed(0)
work = {}


def sync():
    torch.cuda.synchronize()


# ---- distance GEMM + running minimum (features.py:186-190,227): l2_min_pp3_kernel alone, at the shapes it runs at
def l2_rows():
    from cmdiad_amd import engine as eng
    n_launch, seq = 7, 0
    Qmax, live = B * 3136, 54401              # bench.py: 100 352 rows per step, 54 401 of them distinct at its clouds
    q16, _, qsq = ops.normalize_cast(torch.randn(Qmax, D, generator=g).to(DEV), want_f32=False)
    xyz = eng.Bank(torch.randn(76518, D, generator=g).to(DEV))
    rgb = eng.Bank(torch.randn(19129, D, generator=g).to(DEV))
    shard = eng.Bank(xyz.f32, 0, 8)           # rank 0's row shard of an 8-rank node (9 600 rows)
    cnt = torch.tensor([live], dtype=torch.int32, device=DEV)
    cap = (live + 255) // 256 * 256
    seg_cnt = torch.full((8,), live, dtype=torch.int32, device=DEV)
    q_seg, s_seg = q16[:cap].repeat(8, 1), qsq[:cap].repeat(8)
    k_all, k_seg = ops.new_keys(Qmax, DEV, runner=True), ops.new_keys(8 * cap, DEV, runner=True)   # best + runner-up, as the pipeline launches it
    shapes = (
        ("bench", live, xyz, lambda: ops.l2_min_keys_counted(q16, qsq, cnt, xyz.bf16, xyz.sqnorm, k_all, 0),
         "the bench's launch: 54 401 live rows (device-resident count, grid sized for 100 352) x xyz library 76 518 (+26 pad) x 768, " + ("bf16" if xyz.bf16.dtype == torch.bfloat16 else "fp16") + " operands"),
        ("all", Qmax, xyz, lambda: ops.l2_min_keys(q16, qsq, xyz.bf16, xyz.sqnorm, k_all, 0),
         "every row searched (CMDIAD_DEDUP=0, the reference's cdist): 100 352 x 76 518 x 768"),
        ("w8", 8 * live, shard, lambda: ops.l2_min_keys_segments(q_seg, s_seg, seg_cnt, cap, shard.bf16, shard.sqnorm, k_seg, 0),
         "one rank of an 8-rank node (configs[3]): 8 segments x 54 401 live rows x its 9 600-row shard, one segments launch"),
        ("rgb", B * 784, rgb, lambda: ops.l2_min_keys(q16[:B * 784], qsq[:B * 784], rgb.bf16, rgb.sqnorm, k_all[:, :B * 784].contiguous(), 0),
         "rgb library: 25 088 x 19 129 (+71 pad) x 768"),
    )
    n_warm = 20     # ~0.2 s of back-to-back launches first: whatever a process measures first loses to the clocks settling (r4_notes 16)
    for _ in range(n_warm):
        shapes[1][3]()
    sync()
    seq = n_warm
    for tag, rows_q, bank, fn, what in shapes:
        for _ in range(n_launch):
            fn()
            sync()
        nb = bank.bf16.shape[0]
        work[f"l2_min_pp3_kernel/{tag}"] = dict(flops=2.0 * rows_q * nb * D, bytes=(nb + rows_q) * D * 2 + 12 * rows_q, what=what,
                                               seq=[seq, n_launch])
        seq += n_launch


if not only_hbm:
    l2_rows()
# ---- re-weighting scan (features.py:235-254): 32 probes, the fp32 library streamed once
for name, rows in (() if only_l2 else (("xyz", 76518), ("rgb", 19129))):
    bank = torch.randn(rows, D, generator=g).to(DEV)
    blk = ops.bank_block16(bank)
    probes = bank[:32].contiguous()
    for _ in range(30):
        ops.reweight_scan(probes, bank, blk)
    sync()
    work[f"reweight_scan_mfma_kernel/{rows}"] = dict(bytes=rows * D * 4, what=f"re-weighting scan, {name} library {rows} x 768 fp32, 32 probes",
                                                    seq=[0 if name == "xyz" else 30, 30])
    del bank, blk
# ---- the same two scans as ONE launch pair (cmdiad_reweight_scan_pair: what a scored batch runs)
if not only_l2:
    ba, bb = torch.randn(76518, D, generator=g).to(DEV), torch.randn(19129, D, generator=g).to(DEV)
    ka, kb = ops.bank_block16(ba), ops.bank_block16(bb)
    pa, pb_ = ba[:32].contiguous(), bb[:32].contiguous()
    for _ in range(30):
        ops.reweight_scan_pair(pa, ba, ka, pb_, bb, kb)
    sync()
    work["reweight_scan_mfma_kernel/pair"] = dict(bytes=(76518 + 19129) * D * 4, what="re-weighting scans of BOTH libraries as one launch pair (xyz 76518 + rgb 19129 rows x 768 fp32, 32 probes each)", seq=[60, 30])
    del ba, bb, ka, kb
# ---- greedy coreset round (features.py:401-420): 765 184 x 334 fp16 rows per round
if not only_l2:
    n, d = 765184, 334
    z = torch.randn(n, d, device=DEV)
    coreset.greedy_coreset(z, 5)
    sync()
    coreset.greedy_coreset(z, 201)
    sync()
    work["coreset_round_kernel"] = dict(bytes=n * d * 2, what="greedy coreset round, 765 184 x 334 fp16 (bagel xyz)")
    del z
if not only_hbm and not only_l2:
    # ---- farthest point sampling + kNN grouping (models/models.py:70-113)
    pcs = torch.cat([synth_cloud_fixed_n(1000 + i, 24576) for i in range(B)]).to(DEV)
    xyz, nz, pix2pt, nv = ops.unorganize(pcs, 24576)
    for _ in range(6):
        idx, cen = ops.fps(xyz, G, nv)
    sync()
    work["fps_ragged_kernel/fixed"] = dict(evals=B * G * 24576, bytes=B * (24576 * 12 + G * 16), what="FPS, 32 clouds x 24 576 points, 1024 samples (latency-bound: distance evaluations/s)")
    for _ in range(6):
        ops.knn_group(xyz, cen, K, nv)
    sync()
    work["knn_grid_query_kernel"] = dict(evals=B * G * 24576, bytes=B * (24576 * 12 + G * 12 + G * K * 20), what="kNN grouping, 1024 centres x 128 neighbours per cloud: the query half of the neighbourhood search (rate in brute-force-equivalent evaluations: 805 M would be evaluated by the streaming kernel, ~16 M are)")
    work["knn_grid_build_kernel"] = dict(bytes=B * 24576 * (12 + 16), what="kNN grouping: the cloud binned into a 64 x 64 grid (counting sort, one workgroup per cloud)")
    idx3, w3 = None, None
    for _ in range(6):
        idx3, w3 = ops.interp3nn(xyz, cen, nv)
    sync()
    work["interp3nn_grid_kernel"] = dict(evals=B * G * 24576, bytes=B * 24576 * (12 + 24), what="3-NN + weights, 24 576 points x 1024 centres per cloud, neighbourhood search on binned centres (brute-force-equivalent evaluations)")
    import numpy as np
    rs = np.random.RandomState(8)
    fr = (0.35 + 0.30 * rs.rand(B)) / 0.85
    pcs_v = torch.cat([synth_cloud(7000 + i, float(fr[i])) for i in range(B)]).to(DEV)
    xyz_v, _, _, nv_v = ops.unorganize(pcs_v, None)
    for _ in range(6):
        ops.fps(xyz_v, G, nv_v)
    sync()
    work["fps_ragged_kernel/var"] = dict(evals=int(G * nv_v.sum().item()), what=f"FPS, ragged batch ({int(nv_v.min())} ... {int(nv_v.max())} points, padded {xyz_v.shape[1]})")
    # ---- Point-MAE encoder (models/models.py:200-215)
    from oracle import nets  # noqa: E402  (synthetic weights only)
    w = fold_pointmae_encoder(nets.synth_state_dict("pointmae", 21), "encoder.", DEV)
    nb = (torch.randn(B * G * K, 3, generator=g) * 0.01).to(DEV)
    for _ in range(6):
        h2, _, g16 = ops.encoder_stage1(nb, w["w1b1"], w["W2"], w["b2"], B * G, K)
        gb, _ = ops.gemm(g16, w["W3a"], bias=w["b3"], want_f32=True, want_bf16=False)
        ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], B * G, K)
    sync()
    pts = B * G * K
    work["encoder_stage1_persist_kernel"] = dict(flops=2.0 * pts * (3 * 128 + 128 * 256), bytes=pts * (12 + 512), what="encoder stage 1: conv1 + conv2 + group max, h2 written as bf16")
    work["encoder_tail_persist_kernel"] = dict(flops=2.0 * pts * (256 * 512 + 512 * 384), bytes=pts * 512, what="encoder tail: conv3 (per-point half) + ReLU + conv4 + group max")
    del nb, h2
    # ---- attention at the two production shapes (models/models.py:148-160 and timm's blocks, models.py:48): q pre-scaled as gemm_qkv does
    seq_a = 0
    for tag, T, H, what in (("vit", 785, 12, "ViT-B/8: 785 tokens x 12 heads x 64"), ("pmae", 1024, 6, "Point-MAE: 1 024 tokens x 6 heads x 64")):
        Tp = (T + 63) // 64 * 64
        qa = (torch.randn(B, H, Tp, 64, generator=g) * 0.18).to(DEV).bfloat16()
        ka_ = torch.randn(B, H, Tp, 64, generator=g).to(DEV).bfloat16()
        va = torch.randn(B, H, 64, Tp, generator=g).to(DEV).bfloat16()
        for _ in range(8):
            ops.attention(qa, ka_, va, B, H, T)
        sync()
        work[f"attention_kernel/{tag}"] = dict(flops=4.0 * B * H * T * T * 64, what=f"fused softmax(q k^T) v, batch 32, {what} (useful FLOPs: padding not counted)", seq=[seq_a, 8])
        seq_a += 8
    # ---- implicit-GEMM convolution of the distillation heads (hallucination_network.py:72-143)
    x = torch.randn(B, 56, 56, 768, generator=g).to(DEV).bfloat16()
    wc = (torch.randn(768, 9 * 768, generator=g) / (9 * 768) ** 0.5).to(DEV).bfloat16()
    out = torch.empty(B, 56, 56, 768, device=DEV, dtype=torch.bfloat16)
    for _ in range(6):
        ops.conv2d_nhwc(x, wc, 768, 3, 1, act=ops.ACT_RELU, out_bf16=out, want_bf16=False)
    sync()
    work["conv_igemm_kernel"] = dict(flops=2.0 * B * 3136 * 768 * 9 * 768, what="3x3 convolution 768 -> 768 on the 56 x 56 token grid, batch 32")
    del x, out
    # ---- weight-gradient product of the trainer (pretrain.py:148-154)
    M = B * 3136
    P = torch.randn(M, 1920, generator=g).to(DEV).bfloat16()
    Q = torch.randn(M, 1920, generator=g).to(DEV).bfloat16()
    for _ in range(6):
        ops.gemm_tn(P, Q, split_k=8, want_colsum=True)
    sync()
    work["gemm_tn_kernel"] = dict(flops=2.0 * M * 1920 * 1920, what="dW = sum_m P[m,:]^T Q[m,:], 1920 x 1920 x 100 352, split-K 8")
out_dir = os.environ.get("STANDALONE_WORK_DIR", "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
if not only_hbm and not only_l2:      # (the counter passes repeat a subset: they must not overwrite the full table)
    json.dump(work, open(os.path.join(out_dir, "standalone_work.json"), "w"), indent=1)
elif only_l2:
    json.dump(work, open(os.path.join(out_dir, "standalone_work_l2.json"), "w"), indent=1)
print("done", sorted(work))
