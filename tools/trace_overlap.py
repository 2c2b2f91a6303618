#!/usr/bin/env python3
"""How the streams of the pipeline share the chip, from a rocprofv3 kernel trace of bench.py:
    python tools/trace_overlap.py gpurun_out/prof_r4/trace/*/*kernel_trace.csv
Over eight steady-state steps (vit_assemble launches 4 .. 12): the share of time with 0 / 1 / 2 / 3+ kernels in flight, the time in
which everything in flight together has fewer workgroups than the chip has CUs, and which kernels run alone."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0][:46], g // w, r["Stream_Id"]))
ev.sort()
va = [e for e in ev if "vit_assemble" in e[2]]
a, b, steps = va[4][0], va[12][0], 8
print(f"window {(b - a) / 1e6:.2f} ms = {(b - a) / 1e6 / steps:.2f} ms/step")
pts = []
for i, (s, e, n, bl, st) in enumerate(ev):
    s2, e2 = max(s, a), min(e, b)
    if e2 > s2:
        pts.append((s2, 0, i)); pts.append((e2, -1, i))
pts.sort(key=lambda x: (x[0], x[1]))
act, last = set(), a
hist, alone, low = collections.Counter(), collections.Counter(), 0
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        hist[min(len(act), 3)] += dt
        if act and sum(ev[j][3] for j in act) < 256:
            low += dt
        if len(act) == 1:
            alone[ev[next(iter(act))][2]] += dt
    last = t
    if d == 0: act.add(i)
    else: act.discard(i)
tot = b - a
for k in sorted(hist):
    print(f"kernels in flight {k}{'+' if k == 3 else ''}: {100 * hist[k] / tot:.1f} %")
print(f"everything in flight together < 256 workgroups: {100 * low / tot:.1f} % ({low / 1e6 / steps:.2f} ms/step)")
print("running alone, ms/step:")
for n, v in alone.most_common(12):
    print(f"   {n:46s} {v / 1e6 / steps:.3f}")
busy = collections.Counter()
for s, e, n, bl, st in ev:
    s2, e2 = max(s, a), min(e, b)
    if e2 > s2: busy[st] += e2 - s2
for st, v in busy.most_common():
    print(f"stream {st}: kernels in flight {v / 1e6 / steps:.2f} ms/step")
