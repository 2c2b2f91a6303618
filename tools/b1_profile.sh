#!/bin/bash
# Kernel trace of the B = 1 drop-in (tools/dropin_bench.py): gpurun_out/prof_b1/
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_b1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/dropin_bench.py > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log"
python3 tools/summarize_profile.py "$OUT"/*/*kernel_trace.csv 45 | head -40
