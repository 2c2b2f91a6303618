export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_b1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 tools/dropin_bench.py 256 64 > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log"
find "$OUT" -name "*.db" -delete
python3 tools/trace_overlap.py "$OUT"/*/*kernel_trace.csv | head -30
