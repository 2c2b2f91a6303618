#!/bin/bash
# L2-side request counters of the distance GEMM per tile variant (CMDIAD_L2_TILE): does a formulation re-request lines?
OUT=$PWD/gpurun_out/l2tcc; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
for t in ${1:-"5 6"}; do
  export CMDIAD_L2_TILE=$t
  rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/t$t" -- python3 tools/l2_one.py > "$OUT/t$t.log" 2>&1
  python3 - "$OUT/t$t" $t <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "l2_min_pp" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("tile", sys.argv[2], {k: f"{sum(v) / len(v):.4g}" for k, v in agg.items()}, flush=True)
PY
done
rm -rf "$OUT"/t*/
