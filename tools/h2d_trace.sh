#!/bin/bash
# Kernel + memory-copy trace of the H2D-fed steps (bench.py's h2d_inclusive leg): where the pinned-host -> device copies of the next
# batch sit relative to the kernels of the current step (no counters in this run: tracing domains only).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=$PWD/gpurun_out/h2d_trace
rm -rf "$O"; mkdir -p "$O"
export CMDIAD_BENCH_LEGS=h2d_inclusive
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/t" -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 3 > "$O/bench.log" 2>&1
tail -c 600 "$O/bench.log"
find "$O" -name "*.db" -delete
ls -la "$O"/t/*/ | head
python3 tools/h2d_trace_summary.py "$O"/t/*/*memory_copy_trace.csv "$O"/t/*/*kernel_trace.csv
