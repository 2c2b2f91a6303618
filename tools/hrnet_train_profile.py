#!/usr/bin/env python3
"""HRNet head training steps alone (for a kernel trace): rocprofv3 --kernel-trace --stats -- python3 tools/hrnet_train_profile.py [batch]"""
import os, sys, time, torch
os.environ.setdefault("CMDIAD_HRNET_TRAIN", "hip")
sys.path.insert(0, os.getcwd())
from cmdiad_amd.models.hrnet import HRNet
from oracle import heads
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gen = torch.Generator().manual_seed(1)
img, feat = torch.randn(B, 3, 224, 224, generator=gen).cuda(), torch.randn(B, 3136, 768, generator=gen).cuda()
m = HRNet(512, 768, 0.1); m.load_state_dict(heads.synth_head_state_dict("hrnet", 41)); m.cuda().train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(); loss = m(img, feat); loss.backward(); opt.step(); return loss
W, K = int(os.environ.get("WARM", "1")), int(os.environ.get("STEPS", "3"))   # (the first steps grow the caching allocator: WARM=5 for timings)
for _ in range(W): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); print(f"hrnet training step, batch {B}, {os.environ['CMDIAD_HRNET_TRAIN']}: {(time.perf_counter() - t0) / K * 1e3:.1f} ms")
