#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count and mean duration -- and, because the pipeline runs
its stages on several streams, the same figures split into the launches that ran ALONE on the chip (another kernel in flight for
less than 5 % of the launch) and the ones that SHARED it: a kernel's duration beside another stream's kernels contains the time it
spends without the CUs, so only the isolated mean is a kernel property (round 4's one mean of the dominant kernel mixed 12
overlapped launches of 9.4 ms with 6 isolated ones of 5.7 ms).
usage: tools/summarize_profile.py <kernel_trace.csv> <steps_in_run> > profiles/rN_summary.md"""
import bisect
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# timeline of "kernels in flight": breakpoints ts[], busy2[i] = ns in [ts[0], ts[i]) with >= 2 kernels in flight
ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
ts, busy2, n, acc = [], [], 0, 0
for i, (t, d) in enumerate(ev):
    if ts:
        acc += (t - ts[-1]) if n >= 2 else 0
    ts.append(t)
    busy2.append(acc)
    n += d


def shared_ns(s, e):
    """ns of [s, e) during which at least one OTHER kernel was in flight"""
    def upto(t):
        i = bisect.bisect_right(ts, t) - 1
        return busy2[i] + 0 if i < 0 else busy2[i]      # t is itself a breakpoint (every start / end is)
    return upto(e) - upto(s)


agg = collections.defaultdict(lambda: {"all": [], "alone": [], "shared": []})
for r, (s, e) in zip(rows, iv):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    us = (e - s) / 1e3
    a = agg[(name, r["Grid_Size_X"], r["Grid_Size_Y"])]
    a["all"].append(us)
    a["alone" if shared_ns(s, e) < 0.05 * max(e - s, 1) else "shared"].append(us)
tot = sum(sum(v["all"]) for v in agg.values())


def mean(v):
    return f"{sum(v) / len(v):.1f}" if v else "-"


print("| kernel | grid (threads) | launches/step | mean us | ms/step | share | alone: launches, mean us | beside other kernels: launches, mean us |")
print("|---|---|---|---|---|---|---|---|")
for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -sum(kv[1]["all"])):
    v = a["all"]
    if sum(v) / tot < 0.001:
        continue
    print(f"| `{name[:60]}` | {gx}x{gy} | {len(v) / steps:.1f} | {mean(v)} | {sum(v) / steps / 1e3:.3f} | {100 * sum(v) / tot:.1f}% | "
          f"{len(a['alone'])}, {mean(a['alone'])} | {len(a['shared'])}, {mean(a['shared'])} |")
print(f"\nGPU kernel time per step: {tot / steps / 1e3:.2f} ms ({steps} steps incl. warm-up); 'alone' = another kernel in flight for < 5 % of the launch")
