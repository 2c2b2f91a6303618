#!/usr/bin/env python3
"""Conv FtoF head training steps alone, hand-written path (for a kernel trace):
rocprofv3 --kernel-trace --stats -- python3 tools/conv_head_profile.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd.models import hallucination_network as hn
from oracle import heads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gen = torch.Generator().manual_seed(1)
a, b = torch.randn(B, 3136, 768, generator=gen).cuda(), torch.randn(B, 3136, 768, generator=gen).cuda()
m = hn.HallucinationCrossModalityConv(None, 768, 768); m.load_state_dict(heads.synth_head_state_dict("conv_ftof", 41)); m.cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(); lx, lr = m(a, b, False, "l2"); (lx + lr).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"conv FtoF head training step, batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
