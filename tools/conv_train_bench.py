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
img = torch.randn(B, 3, 224, 224, generator=gen).cuda()
for kind, make, call in (("ftoi_conv", lambda: hn.HallucinationFeatureToInputConv(None, 768), lambda m: m(a, img)),
                         ("hrnet", lambda: __import__("cmdiad_amd.models.hrnet", fromlist=["HRNet"]).HRNet(512, 768, 0.1), lambda m: m(img, a)),
                         ("ftoi_mlp", lambda: hn.HallucinationRGBFeatureToXYZInputMLP(__import__("types").SimpleNamespace(estimate_depth=False), 768), lambda m: m(a, img))):
    for mode in ("hip", "torch"):
        os.environ["CMDIAD_CONV_TRAIN"] = os.environ["CMDIAD_HRNET_TRAIN"] = mode
        m = make(); m.load_state_dict(heads.synth_head_state_dict(kind, 41)); m.cuda().train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        def step():
            opt.zero_grad(); loss = call(m); loss.backward(); opt.step(); return float(loss.detach())
        step(); step(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): last = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"{kind} head training step, batch {B}, {mode}: {dt * 1e3:.1f} ms (loss {last:.2f}), peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
        del m, opt; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
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
