#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count and mean duration.
usage: tools/summarize_profile.py <kernel_trace.csv> <steps_in_run> > profiles/rN_summary.md"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    agg[(name, r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"| kernel | grid (threads) | launches/step | mean us | ms/step | share |\n|---|---|---|---|---|---|")
for (name, gx, gy), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.001:
        continue
    print(f"| `{name[:60]}` | {gx}x{gy} | {len(v) / steps:.1f} | {sum(v) / len(v):.1f} | {sum(v) / steps / 1e3:.3f} | {100 * sum(v) / tot:.1f}% |")
print(f"\nGPU kernel time per step: {tot / steps / 1e3:.2f} ms ({steps} steps incl. warm-up)")
