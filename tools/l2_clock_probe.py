#!/usr/bin/env python3
"""The distance GEMM's own shader clock and power: the bench's launch (54 401 live rows x 76 544 x 768) in a loop for a few seconds
per operand type while rocm-smi (read-only) samples sclk / power twice a second -- what `roofline.frac` (against the NOMINAL
2.5 PFLOP/s at 2 400 MHz) is in terms of the clock the chip actually holds under its power cap.  python tools/l2_clock_probe.py [s]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.getcwd())
from cmdiad_amd import ops  # noqa: E402

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
Qmax, Nb, D, live = 100352, 76544, 768, 54401


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


g = torch.Generator().manual_seed(0)
bank = torch.randn(Nb, D, generator=g).cuda()
qq = torch.randn(Qmax, D, generator=g).cuda()
cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
for name, b, q in (("bf16 random operands", bank, qq), ("fp16 random operands", bank, qq), ("bf16 all-zero operands", torch.zeros_like(bank), torch.zeros_like(qq))):
    dt = torch.float16 if name.startswith("fp16") else torch.bfloat16
    b16, _, bsq = ops.normalize_cast(b, dtype=dt)
    q16, _, qsq = ops.normalize_cast(q, dtype=dt)
    keys = ops.new_keys(Qmax, "cuda", runner=True)
    for _ in range(3):
        ops.l2_min_keys_counted(q16, qsq, cnt, b16, bsq, keys)
    torch.cuda.synchronize()
    samples, stop = [], False

    def watch():
        while not stop:
            samples.append(smi())
            time.sleep(0.4)
    th = threading.Thread(target=watch)
    th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.perf_counter(), 0
    e0.record()
    while time.perf_counter() - t0 < SECS:
        for _ in range(10):
            ops.l2_min_keys_counted(q16, qsq, cnt, b16, bsq, keys)
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / n
    sc = [s for s, _ in samples[1:] if s]
    pw = [p for _, p in samples[1:] if p]
    tf = 2.0 * live * Nb * D / ms / 1e9
    clk = sum(sc) / max(len(sc), 1)
    print(f"{name:24s} {ms:6.3f} ms  {tf:6.0f} TFLOP/s = {tf / 2500:.3f} of the nominal peak; sclk {clk:5.0f} MHz ({min(sc, default=0):.0f}-{max(sc, default=0):.0f}), "
          f"power {sum(pw) / max(len(pw), 1):5.0f} W  ->  {tf / (2500 * clk / 2400):.3f} of the peak at that clock", flush=True)
