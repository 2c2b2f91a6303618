#!/usr/bin/env python3
"""ViT-B/8 forward alone at B=32 (for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/vit_profile.py)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cmdiad_amd import runtime
from oracle import nets
which = sys.argv[1] if len(sys.argv) > 1 else "vit"
if which == "vit":
    m = runtime.PackedViT(nets.synth_state_dict("vit", 31), device="cuda")
    x = torch.randn(32, 3, 224, 224, device="cuda")
    f = lambda: m.forward_tokens(x)
else:
    m = runtime.PackedPointMAE(nets.synth_state_dict("pointmae", 21), device="cuda")
    tok = torch.randn(32 * 1024, 384, device="cuda"); cen = torch.randn(32, 1024, 3, device="cuda")
    f = lambda: m.transform(tok.clone(), cen)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"{which}: {e0.elapsed_time(e1) / 10:.3f} ms per forward")
