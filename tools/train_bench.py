#!/usr/bin/env python3
"""BASELINE configs[2]: FtoF distillation training step (both directions, forward + backward + Adam) on
[32, 3136, 1536] feature batches.  Not the headline bench (bench.py); reports

  * steps/s and TFLOP/s (7.99 TFLOP per step) with the batch resident in HBM,
  * the same with batches streamed from .pt files through cmdiad_amd.dataset.FeatureRing,
    i.e. whether the loader is off the critical path.

    python tools/train_bench.py [--steps 30] [--warmup 5] [--files 96] [--dir /tmp/cmdiad_feats]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmdiad_amd import train  # noqa: E402
from cmdiad_amd.dataset import FeatureRing  # noqa: E402
from cmdiad_amd.models.hallucination_network import HallucinationCrossModalityNetwork  # noqa: E402
from cmdiad_amd.utils import lr_sched  # noqa: E402

STEP_TFLOP = 7.99


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--files", type=int, default=96)
    ap.add_argument("--dir", default="/tmp/cmdiad_feats")
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(3407)
    net = HallucinationCrossModalityNetwork(None, 768, 768).to(dev)
    opt = train.FusedAdam(net.parameters(), lr=5e-4)
    sched = type("A", (), dict(lr=5e-4, warmup_epochs=10))()

    def step(x, it):
        lr_sched.adjust_learning_rate(opt, it / 100.0, sched)
        lx, lr_ = net(x[:, :, :768], x[:, :, 768:], False, "l2")
        opt.zero_grad(set_to_none=True)
        (lx + lr_).backward()
        opt.step()

    out = {"config": "configs[2]: FtoF distillation, batch 32 x 3136 tokens x (768 + 768), l2 loss, fused Adam"}
    # ---- resident batch
    x = torch.randn(32, 3136, 1536, device=dev)
    for i in range(a.warmup):
        step(x, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(x, i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out["resident"] = {"ms_per_step": round(dt * 1e3, 2), "steps_per_s": round(1 / dt, 2), "TFLOPs": round(STEP_TFLOP / dt, 1)}
    # ---- streamed from disk through the ring
    os.makedirs(a.dir, exist_ok=True)
    g = torch.Generator().manual_seed(1)
    for i in range(a.files):
        p = os.path.join(a.dir, f"synthetic{i}.pt")
        if not os.path.exists(p):
            torch.save(torch.randn(3136, 1536, generator=g), p)
    ring = FeatureRing(a.dir, 32, shuffle=True, drop_last=True, device=dev, depth=3, readers=8)
    first = time.perf_counter()
    for xb, _ in ring:          # epoch 1: from disk, fills the HBM-resident cache
        step(xb, 0)
    torch.cuda.synchronize()
    first = (time.perf_counter() - first) / max(len(ring), 1)
    n, t0 = 0, None
    while n < a.warmup + a.steps:
        for xb, _ in ring:      # later epochs: on-device gather
            if n == a.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            step(xb, n)
            n += 1
            if n == a.warmup + a.steps:
                break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out["ring"] = {"first_epoch_ms_per_step_from_disk": round(first * 1e3, 1), "ms_per_step": round(dt * 1e3, 2), "steps_per_s": round(1 / dt, 2), "TFLOPs": round(STEP_TFLOP / dt, 1),
                             "files": a.files, "MB_per_step": round(32 * 3136 * 1536 * 4 / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
