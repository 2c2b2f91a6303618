#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes of `bench.py` into per-kernel, per-launch numbers.

    python tools/pmc_summary.py gpurun_out/pmc_r1 > profiles/r1_pmc.md      (also writes profiles/r1_pmc.json)

Expects one sub-directory per pass under the given directory (fetch/, write/, l2/, sq/ ...), each holding
rocprofv3's *_counter_collection.csv.  Corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM:
FETCH_SIZE and WRITE_SIZE are in KiB of fabric-side (L2 <-> Infinity Cache/HBM) requests; on gfx950
FETCH_SIZE tallies the 128-byte requests of 16-B-per-lane streaming reads (global_load and LDS-DMA) at
64 bytes, so it is DOUBLED; WRITE_SIZE is taken as reported.  Infinity-Cache hits are included in both.
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:72]


def load(root):
    """{(kernel, grid): {counter: [values per launch]}}"""
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            data[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return data


def main():
    root = sys.argv[1]
    out_json = sys.argv[2] if len(sys.argv) > 2 else os.path.join("profiles", "r1_pmc.json")
    data = load(root)
    rows = []
    for (k, grid), c in data.items():
        mean = {n: sum(v) / len(v) for n, v in c.items()}
        row = {"kernel": k, "grid_threads": grid, "launches": max(len(v) for v in c.values())}
        if "FETCH_SIZE" in mean:
            row["fetch_bytes"] = 2.0 * 1024 * mean["FETCH_SIZE"]  # gfx950 correction x2
        if "WRITE_SIZE" in mean:
            row["write_bytes"] = 1024 * mean["WRITE_SIZE"]
        if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
            row["l2_hit"] = mean["TCC_HIT_sum"] / max(mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"], 1.0)
        if "SQ_WAVE_CYCLES" in mean:
            wc = max(mean["SQ_WAVE_CYCLES"], 1.0)
            for n, key in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_ANY", "issuing")):
                if n in mean:
                    row[key] = mean[n] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "SQ_BUSY_CU_CYCLES" in mean:
            row["mfma_busy"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / max(mean["SQ_BUSY_CU_CYCLES"], 1.0)
        if "SQ_LDS_BANK_CONFLICT" in mean and "SQ_LDS_IDX_ACTIVE" in mean:
            row["lds_conflict"] = mean["SQ_LDS_BANK_CONFLICT"] / max(mean["SQ_LDS_IDX_ACTIVE"], 1.0)
        rows.append(row)
    rows.sort(key=lambda r: -(r.get("fetch_bytes", 0) + r.get("write_bytes", 0)))
    json.dump(rows, open(out_json, "w"), indent=1)
    cols = ["fetch_bytes", "write_bytes", "l2_hit", "parked", "issue_stall", "issuing", "mfma_busy", "lds_conflict"]
    print("| kernel | grid | launches | fetch MB (x2 corrected) | write MB | L2 hit | parked | issue-stall | issuing | MFMA_BUSY/BUSY_CU (raw, 4 SIMDs per CU) | LDS conflict |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:40]:
        f = lambda k, s=1.0, fmt="{:.1f}": fmt.format(r[k] / s) if k in r else "-"
        print(f"| `{r['kernel']}` | {r['grid_threads']} | {r['launches']} | {f('fetch_bytes', 1e6)} | {f('write_bytes', 1e6)} | "
              f"{f('l2_hit', 1, '{:.3f}')} | {f('parked', 1, '{:.2f}')} | {f('issue_stall', 1, '{:.2f}')} | {f('issuing', 1, '{:.2f}')} | "
              f"{f('mfma_busy', 1, '{:.2f}')} | {f('lds_conflict', 1, '{:.3f}')} |")


if __name__ == "__main__":
    main()
