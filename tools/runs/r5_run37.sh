#!/bin/bash
# round 5, GPU call 37: the stand-alone kernel trace re-taken on the final tree (counter passes of run 18 stay: their kernels did not change)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r5_final
rm -rf "$OUT"; mkdir -p "$OUT"
export STANDALONE_WORK_DIR="$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/standalone" -- python3 tools/standalone_kernels.py > "$OUT/standalone.log" 2>&1
echo "standalone trace rc=$?"
du -sh "$OUT"
