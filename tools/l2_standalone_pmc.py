#!/usr/bin/env python3
"""The dominant kernel's counters in the regime `roofline.frac` is quoted in -- the launch ALONE on the chip:
    python tools/l2_standalone_pmc.py gpurun_out/prof_r5 profiles/r5
reads the stand-alone counter passes of tools/profile_round.sh (<prof>/standalone_l2_pmc/<pass>/ ..., tools/standalone_kernels.py l2:
20 warm-up launches, then 7 launches each of the four shapes) and writes profiles/r5_l2_standalone.json (one row per shape: fabric
bytes per launch -- 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE --, L2 hit rate, parked / stalled /
issuing shares, MFMA busy, LDS conflict rate) and profiles/r5_pmc_meta.json (sha256 of the kernel's sources: bench.py emits
`roofline.traffic` only while they are unchanged)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

prof, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WARM, PER = 20, 7
SHAPES = ("bench", "all", "w8", "rgb")
vals = collections.defaultdict(lambda: collections.defaultdict(list))      # shape -> counter -> values
for f in glob.glob(os.path.join(prof, "standalone_l2_pmc", "**", "*counter_collection.csv"), recursive=True):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "l2_min_pp3_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in rows.items():
        v.sort()
        v = [x for _, x in v][WARM:]
        for i, shape in enumerate(SHAPES):
            vals[shape][c] += v[i * PER + 1:(i + 1) * PER]          # (the first launch of a shape is dropped)
res = []
for shape in SHAPES:
    m = {c: sum(v) / len(v) for c, v in vals[shape].items() if v}
    row = {"kernel": "l2_min_pp3_kernel", "shape": shape, "regime": "stand-alone launch (tools/standalone_kernels.py l2)"}
    if "FETCH_SIZE" in m:
        row["fetch_bytes"] = 2.0 * 1024 * m["FETCH_SIZE"]
    if "WRITE_SIZE" in m:
        row["write_bytes"] = 1024 * m["WRITE_SIZE"]
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        row["l2_hit"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0)
    if "SQ_WAVE_CYCLES" in m:
        wc = max(m["SQ_WAVE_CYCLES"], 1.0)
        for n, key in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_ANY", "issuing")):
            if n in m:
                row[key] = m[n] / wc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
        row["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(m["SQ_BUSY_CU_CYCLES"], 1.0)
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        row["lds_conflict"] = m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0)
    res.append(row)
json.dump(res, open(out + "_l2_standalone.json", "w"), indent=1)
sources = ["cmdiad_amd/csrc/l2min.hip", "cmdiad_amd/csrc/gemm_core.h"]
h = hashlib.sha256()
for f in sources:
    h.update(open(os.path.join(ROOT, f), "rb").read())
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
tag = os.path.basename(out)
json.dump({"sources": sources, "sha256": h.hexdigest(), "commit": commit, "standalone": f"profiles/{tag}_l2_standalone.json",
           "note": f"kernel sources of l2_min_pp3_kernel at the time the counter passes of profiles/{tag}_pmc.json / {tag}_l2_standalone.json were taken "
                   f"(tools/profile_round.sh {tag})"}, open(out + "_pmc_meta.json", "w"), indent=1)
for r in res:
    print(json.dumps(r))
