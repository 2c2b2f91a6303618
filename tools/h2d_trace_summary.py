#!/usr/bin/env python3
"""Summary of tools/h2d_trace.sh: every large host->device copy with its duration, rate, and what ran beside it."""
import csv
import sys

copies = list(csv.DictReader(open(sys.argv[1])))
kernels = list(csv.DictReader(open(sys.argv[2])))
print("copy columns:", list(copies[0].keys()) if copies else None)
ks = sorted(((int(k["Start_Timestamp"]), int(k["End_Timestamp"]), k["Kernel_Name"]) for k in kernels))
big = []
for c in copies:
    s, e = int(c["Start_Timestamp"]), int(c["End_Timestamp"])
    size = int(c.get("Size", c.get("Bytes", 0)) or 0) if (c.get("Size") or c.get("Bytes")) else 0
    d = c.get("Direction", c.get("Kind", ""))
    big.append((s, e, size, d))
big.sort()
t0 = ks[0][0]
kinds = {}
for s, e, size, d in big:
    kinds.setdefault(d, []).append((e - s, size))
for d, v in kinds.items():
    tot = sum(x for x, _ in v)
    print(f"{d}: {len(v)} copies, total {tot / 1e6:.2f} ms, bytes {sum(b for _, b in v) / 1e6:.1f} MB, largest {max(x for x, _ in v) / 1e3:.0f} us")
print("largest host-to-device copies (start ms, duration us, MB, GB/s, kernels overlapping):")
h2d = [x for x in big if "HOST_TO_DEVICE" in x[3].upper() or "H2D" in x[3].upper()]
for s, e, size, d in sorted(h2d, key=lambda x: -(x[1] - x[0]))[:14]:
    over = {}
    for a, b, n in ks:
        if a < e and b > s:
            over[n.split("(")[0][:40]] = over.get(n.split("(")[0][:40], 0) + (min(b, e) - max(a, s))
    top = sorted(over.items(), key=lambda kv: -kv[1])[:4]
    print(f"  {(s - t0) / 1e6:9.3f}  {(e - s) / 1e3:8.0f}  {size / 1e6:7.1f}  {size / max(e - s, 1):6.1f}  " + "; ".join(f"{n} {v / 1e3:.0f}us" for n, v in top))
