#!/bin/bash
# Kernel trace of the FtoF distillation training step (tools/train_bench.py, resident batch): gpurun_out/prof_train/
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_train
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/train_bench.py --steps 10 --warmup 2 --files 32 > "$OUT/run.log" 2>&1
tail -2 "$OUT/run.log"
python3 tools/summarize_profile.py "$OUT"/*/*kernel_trace.csv 25 | head -36   # 12 resident + 1 from disk + 12 from the ring
