#!/usr/bin/env python3
"""profiles/rN_standalone.md from the traces of tools/standalone_kernels.py:
    python tools/standalone_summary.py <kernel-trace dir> [<pmc dir>] [standalone_work.json] > profiles/r4_standalone.md
Per kernel (and grid, where one kernel runs at two sizes): launches, mean duration from rocprofv3's kernel trace, the algorithmic
bytes / FLOPs of one launch (tools/standalone_kernels.py) and the fraction of the peak that bounds it (8 TB/s HBM, 2.5 PFLOP/s
dense 16-bit MFMA; /opt/skills/guides/MI355X_MICROARCH.md); with a PMC directory also the fabric-side bytes per launch
(2 x FETCH_SIZE -- gfx950 tallies the 128-byte requests of 16-B-per-lane loads at 64 bytes -- and WRITE_SIZE, KiB -> bytes)."""
import collections
import csv
import glob
import json
import os
import sys

trace_dir = sys.argv[1]
# counter directories, comma-separated (the HBM-bound kernels' passes and the distance GEMM's FETCH / WRITE passes live apart)
pmc_dirs = [d for d in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if os.path.isdir(d)]
pmc_dir = pmc_dirs or None
work_file = next((a for a in sys.argv[2:] if a.endswith(".json")), os.path.join("gpurun_out", "standalone_work.json"))
work = json.load(open(work_file))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    if name.startswith("_ZN"):              # a name the profiler did not demangle: keep the identifier inside it
        import re
        m = re.search(r"\d+([a-z][a-z0-9_]*kernel)", name)
        name = m.group(1) if m else name
    return name


dur = collections.defaultdict(list)
by_name = collections.defaultdict(list)        # kernel name -> [(start, grid, us)] in launch order: `seq` rows (one name, several shapes)
for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        grid = int(r["Grid_Size_X"]) * max(int(r.get("Grid_Size_Y", 1) or 1), 1)
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[(short(r["Kernel_Name"]), grid)].append(us)
        by_name[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), grid, us))
for v in by_name.values():
    v.sort()
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
pmc_seq = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel name -> counter -> [(dispatch id, value)]
if pmc_dir:
    for f in (f for d in pmc_dirs for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            pmc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pmc_seq[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c in pmc_seq.values():
        for v in c.values():
            v.sort()

print("| kernel | grid (threads) | launches | mean us | algorithmic work per launch | achieved | fraction of peak | fabric bytes per launch (PMC) | what |")
print("|---|---|---|---|---|---|---|---|---|")
for key, spec in work.items():
    kname = key.split("/")[0]
    cands = sorted(((k, v) for k, v in dur.items() if kname in k[0]), key=lambda kv: -kv[0][1])
    seq_pmc = None
    if "seq" in spec and cands:             # launches [first, first + count) of this kernel name, in launch order
        first, count = spec["seq"]
        full = next(n for n in by_name if kname in n)
        rows = by_name[full][first:first + count]
        cands = [((full, rows[0][1]), [us for _, _, us in rows])] if rows else []
        seq_pmc = {cn: [val for _, val in vals[first:first + count]] for cn, vals in pmc_seq.get(full, {}).items()}
    if not cands:
        print(f"| `{kname}` | - | 0 | - | - | - | - | - | {spec['what']} (not in the trace) |")
        continue
    if "/" in key and "seq" not in spec:    # the same kernel at two sizes: xyz / rgb library, fixed / ragged batch
        tag = key.split("/")[1]
        cands = [cands[0]] if tag in ("76518", "fixed") else [cands[-1]]
        if tag in ("fixed", "var"):         # same grid (32 blocks): split by launch order (the fixed-N batch runs first)
            k, v = cands[0]
            half = len(v) // 2
            cands = [(k, v[:half] if tag == "fixed" else v[half:])]
    (k, v) = cands[0]
    v = v[1:] if len(v) > 2 else v           # the first launch of a kernel pays code loading
    us = sum(v) / len(v)
    if "flops" in spec:
        ach, frac, unit = spec["flops"] / us / 1e6, spec["flops"] / us / 1e6 / 2500.0, "TFLOP/s"
        algo = f"{spec['flops'] / 1e9:.1f} GFLOP"
    elif "evals" in spec and ("bytes" not in spec or kname.startswith(("fps", "knn", "interp3nn"))):
        ach, frac, unit = spec["evals"] / us / 1e3, None, "G distance evals/s"
        algo = f"{spec['evals'] / 1e6:.0f} M distance evaluations" + (f", {spec['bytes'] / 1e6:.1f} MB" if "bytes" in spec else "")
    else:
        ach, frac, unit = spec["bytes"] / us / 1e3, spec["bytes"] / us / 1e3 / 8000.0, "GB/s"
        algo = f"{spec['bytes'] / 1e6:.1f} MB"
    fab = "-"
    c = seq_pmc if seq_pmc is not None else pmc.get(k)
    if c:
        parts = []
        if "FETCH_SIZE" in c:
            parts.append(f"fetch {2 * 1024 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) / 1e6:.1f} MB")
        if "WRITE_SIZE" in c:
            parts.append(f"write {1024 * sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) / 1e6:.1f} MB")
        fab = ", ".join(parts)
    print(f"| `{k[0]}` | {k[1]} | {len(v)} | {us:.1f} | {algo} | {ach:.1f} {unit} | {'%.3f' % frac if frac is not None else '-'} | {fab} | {spec['what']} |")
