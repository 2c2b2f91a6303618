"""Diagnostic: cmdiad_attention against float64 models of where P is rounded (tests/test_gpu_nets.py stage-by-stage test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmdiad_amd import ops
from cmdiad_amd.runtime import _QkvBuffers
from oracle.nets_rounded import r16
DEV = "cuda"
for B, T, H, sharp in ((2, 785, 12, 1.0), (2, 1024, 6, 1.0), (2, 785, 12, 6.0)):
    g = torch.Generator().manual_seed(T)
    q, k, vt = _QkvBuffers().get(B, H, T, DEV)
    q[:, :, :T] = (sharp * 0.5 * torch.randn(B, H, T, 64, generator=g)).to(torch.bfloat16).to(DEV)
    k[:, :, :T] = (torch.randn(B, H, T, 64, generator=g)).to(torch.bfloat16).to(DEV)
    vt[:, :, :, :T] = (0.5 * torch.randn(B, H, 64, T, generator=g)).to(torch.bfloat16).to(DEV)
    a = ops.attention(q, k, vt, B, H, T).cpu().double()
    qd, kd, vd = q[:, :, :T].cpu().double(), k[:, :, :T].cpu().double(), vt[:, :, :, :T].transpose(-1, -2).cpu().double()
    sc = qd @ kd.transpose(-2, -1)
    def model(round_p, running):
        m_run = torch.full((B, H, T), -float("inf"), dtype=torch.float64)
        l_run = torch.zeros((B, H, T), dtype=torch.float64)
        o = torch.zeros((B, H, T, 64), dtype=torch.float64)
        mfin = sc.amax(-1)
        for t0 in range(0, T, 64):
            st = sc[..., t0:t0 + 64]
            m_new = torch.maximum(m_run, st.amax(-1)) if running else mfin
            alpha = torch.exp2(m_run - m_new) if running else torch.ones_like(mfin)
            p = torch.exp2(st - m_new[..., None])
            l_run = l_run * alpha + p.sum(-1)
            o = o * alpha[..., None] + (r16(p) if round_p else p) @ vd[:, :, t0:t0 + 64]
            m_run = m_new
        return (o / l_run[..., None]).transpose(1, 2).reshape(B * T, H * 64)
    scale = float(a.abs().mean())
    ref1 = model(True, True)
    dev = (a - r16(ref1)).abs()
    bad = (dev > 1e-3 * scale).nonzero()
    print(f"T={T} sharp={sharp}: deviations from the nearest bf16: max {float(dev.max()) / scale:.2e} of scale, {len(bad)} elements above 1e-3 of scale")
    pk = torch.softmax(sc * 0.6931471805599453, -1).amax(-1)      # [B,H,T] largest attention weight of the query
    for r, c in bad[:6].tolist():
        b, t, h, d = r // T, r % T, c // 64, c % 64
        print(f"   row {r} (b {b} t {t}) col {c} (head {h} d {d}): got {float(a[r, c]):.6e} ref {float(ref1[r, c]):.6e} peak weight {float(pk[b, h, t]):.3f}")
    for name, ref in (("P rounded vs running max", model(True, True)), ("P rounded vs final max", model(True, False)), ("P unrounded", model(False, True))):
        out_r = r16(ref)
        err = (a - ref).abs()
        print(f"T={T} sharp={sharp}: {name}: max|err|/scale {float(err.max()) / scale:.2e}, mean {float(err.mean()) / scale:.2e}, "
              f"not-nearest fraction {float((a != out_r).double().mean()):.2e}")
