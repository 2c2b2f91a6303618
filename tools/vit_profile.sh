#!/bin/bash
export TMPDIR=/tmp
for w in vit pmae; do
OUT=$PWD/gpurun_out/prof_$w; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/vit_profile.py $w > "$OUT/run.log" 2>&1
grep "per forward" "$OUT/run.log"
python3 tools/summarize_profile.py "$OUT"/*/*kernel_trace.csv 13 | head -22
done
