#!/usr/bin/env python3
"""Why encoder_tail_persist_kernel takes 2.8 ms on the pipeline's rows and 2.4 ms on N(0, 1/4) rows.
The kernel has no data-dependent path (no early exit, no branch on a value); what differs is what the matrix cores multiply.
This probe times the SAME launch on inputs that differ only in statistics -- the pipeline's own (h2, group bias) and random
rows whose group bias is shifted so that the share of zeros in h3 = ReLU(h2 . W3b^T + gb) (conv4's A operand, h3 lives in LDS only)
runs from ~100 % to ~0 % -- while rocm-smi (read-only) samples the shader clock and the power twice a second.
    python tools/tail_data_probe.py [seconds per case]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops  # noqa: E402
from cmdiad_amd.runtime import fold_pointmae_encoder  # noqa: E402
from cmdiad_amd.synth import synth_cloud_fixed_n  # noqa: E402
from oracle import nets  # noqa: E402

DEV = "cuda"
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
sd = nets.synth_state_dict("pointmae", 21)
w = fold_pointmae_encoder(sd, "encoder.", DEV)
B, G, K, N = 32, 1024, 128, 24576
groups = B * G


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = pw = None
    for ln in out.splitlines():
        if "sclk" in ln and "(" in ln:
            sclk = float(ln.split("(")[1].split("Mhz")[0])
        if "Power" in ln and ":" in ln:
            try:
                pw = float(ln.split(":")[-1].strip())
            except ValueError:
                pass
    return sclk, pw


def case(name, h2, gb):
    _, h3 = ops.gemm(h2[:1 << 20].contiguous(), w["W3b"], act=ops.ACT_RELU, group_bias=gb[:(1 << 20) // K].contiguous(), group_rows=K)
    zeros = float((h3 == 0).float().mean())
    mag = float(h3.float().abs().mean())
    del h3
    samples, stop = [], False

    def watch():
        while not stop:
            samples.append(smi())
            time.sleep(0.4)
    for _ in range(3):
        ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, K)
    torch.cuda.synchronize()
    th = threading.Thread(target=watch)
    th.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < SECS:
        for _ in range(20):
            ops.encoder_tail(h2, gb, w["W3b"], w["W4"], w["b4"], groups, K)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    sc = [s for s, _ in samples[1:] if s]
    pw = [p for _, p in samples[1:] if p]
    print(f"{name:34s} {ms:6.3f} ms  {2.0 * groups * K * (256 * 512 + 512 * 384) / ms / 1e9:5.0f} TFLOP/s   h3 zeros {zeros:5.3f}  mean |h3| {mag:7.3f}   "
          f"sclk {sum(sc) / max(len(sc), 1):6.0f} MHz ({min(sc, default=0):.0f}-{max(sc, default=0):.0f})  power {sum(pw) / max(len(pw), 1):5.0f} W  [{len(sc)} samples]", flush=True)


# the pipeline's own rows: stage 1 of the encoder on the bench's clouds
xyz, nz, pix2pt, nv = ops.unorganize(torch.cat([synth_cloud_fixed_n(1000 + i, N) for i in range(B)]).to(DEV), N)
idx, cen = ops.fps(xyz, G)
_, nb = ops.knn_group(xyz, cen, K)
h2p, gmax, g16 = ops.encoder_stage1(nb.reshape(-1, 3).contiguous(), w["w1b1"], w["W2"], w["b2"], groups, K)
gbp, _ = ops.gemm(g16, w["W3a"], bias=w["b3"], want_f32=True, want_bf16=False)
g = torch.Generator().manual_seed(0)
h2r = (torch.randn(groups * K, 256, generator=g) * 0.5).to(DEV).bfloat16()
gbr = torch.randn(groups, 512, generator=g).to(DEV)
print(f"pipeline h2: mean |.| {float(h2p.float().abs().mean()):.3f}, std {float(h2p.float().std()):.3f}; group bias mean {float(gbp.mean()):.3f} std {float(gbp.std()):.3f}", flush=True)
for rnd in range(2):
    case("pipeline rows (stage 1's h2, gb)", h2p, gbp)
    case("random rows N(0, 1/4), gb N(0, 1)", h2r, gbr)
    for shift in (-30.0, -6.0, 6.0, 30.0):
        case(f"random rows, gb N(0, 1) {shift:+.0f}", h2r, gbr + shift)
    case("random rows x 8 (large h2), gb N(0, 1)", (h2r.float() * 8).bfloat16(), gbr)
    case("all-zero h2, gb = -1 (h3 = 0)", torch.zeros_like(h2r), torch.full_like(gbr, -1.0))
