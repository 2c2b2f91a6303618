#!/bin/bash
# Distance GEMM: block mapping (query tiles per group x library ranges) against time, fabric-side bytes and in-kernel clock.
#   bash tools/l2_fetch_sweep.sh "4:32 8:32 8:16 16:16 4:16"   -> gpurun_out/l2fetch/summary.txt
# Per configuration: one un-profiled run (time), one --pmc FETCH_SIZE pass, one --pmc GRBM_GUI_ACTIVE pass (clock =
# counter / 8 XCDs / kernel wall time, MI355X_MICROARCH.md 'DVFS give-back').  The program sits directly after `--`.
CFGS=${1:-"4:32 8:32 8:16 16:16 4:16"}
OUT=$PWD/gpurun_out/l2fetch; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for cfg in $CFGS; do
    export CMDIAD_L2_QGROUP=${cfg%%:*} CMDIAD_L2_SPLITS=${cfg#*:}
    python3 tools/l2_one.py >> "$OUT/summary.txt" 2>&1
    for ctr in FETCH_SIZE GRBM_GUI_ACTIVE; do
        rocprofv3 --pmc $ctr --output-format csv -d "$OUT/${cfg/:/_}_$ctr" -- python3 tools/l2_one.py > "$OUT/${cfg/:/_}_$ctr.log" 2>&1
    done
    python3 - "$OUT" "${cfg/:/_}" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys
out, cfg = sys.argv[1], sys.argv[2]
for ctr in ("FETCH_SIZE", "GRBM_GUI_ACTIVE"):
    fs = glob.glob(f"{out}/{cfg}_{ctr}/**/*counter_collection.csv", recursive=True)
    if not fs: print(cfg, ctr, "no csv"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "l2_min_pp3" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    if not rows: print(cfg, ctr, "no rows"); continue
    v = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
    dur = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows) / len(rows)
    if ctr == "FETCH_SIZE": print(f"  {cfg}: FETCH_SIZE {v:.4g} KiB/launch x 2 (gfx950: 128-B requests tallied as 64) = {v * 2048 / 1e9:.2f} GB, {dur / 1e6:.3f} ms profiled")
    else: print(f"  {cfg}: GRBM_GUI_ACTIVE {v:.4g} / 8 / {dur / 1e6:.3f} ms = {v / 8 / dur:.3f} GHz")
PY
done
L2_ZEROS=1 CMDIAD_L2_QGROUP=4 CMDIAD_L2_SPLITS=32 python3 tools/l2_one.py >> "$OUT/summary.txt" 2>&1
rm -rf "$OUT"/*_FETCH_SIZE "$OUT"/*_GRBM_GUI_ACTIVE
cat "$OUT/summary.txt"
