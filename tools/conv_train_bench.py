#!/usr/bin/env python3
"""Training step of the convolutional FtoF head (HallucinationCrossModalityConv, both directions: forward, loss, backward, Adam):
the hand-written path (cmdiad_amd/conv_train.py) against the module's own torch layers (CMDIAD_CONV_TRAIN=torch).
    python tools/conv_train_bench.py [batch] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd.models import hallucination_network as hn
from oracle import heads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
gen = torch.Generator().manual_seed(1)
a, b = torch.randn(B, 3136, 768, generator=gen).cuda(), torch.randn(B, 3136, 768, generator=gen).cuda()
flop = 2 * 4 * 3 * 2.0 * B * 3136 * 768 * 9 * 768 - 2 * 2.0 * B * 3136 * 768 * 9 * 768   # 2 towers x 4 convs x (fwd, dgrad, wgrad), no dgrad for the first
for mode in ("hip", "torch"):
    os.environ["CMDIAD_CONV_TRAIN"] = mode
    m = hn.HallucinationCrossModalityConv(None, 768, 768)
    m.load_state_dict(heads.synth_head_state_dict("conv_ftof", 41))
    m.cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    def step():
        opt.zero_grad()
        lx, lr = m(a, b, False, "l2")
        (lx + lr).backward()
        opt.step()
        return float(lx.detach())
    step(); step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): last = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"conv head training step, batch {B}, {mode}: {dt * 1e3:.1f} ms = {flop / dt / 1e12:.0f} TFLOP/s (loss {last:.4f}), "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del m, opt; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
