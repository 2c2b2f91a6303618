#!/bin/bash
# round 5, GPU call 23: ViT forward alone with and without the LayerNorm fold; kernel traces of both
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_23
mkdir -p $O
for i in 1 2 3; do
  python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/default  /' | tee -a $O/vit.log
  CMDIAD_LN_FOLD=1 python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/LN fold  /' | tee -a $O/vit.log
done
for mode in pmae 1; do
  OUT=$PWD/$O/prof_$mode; rm -rf "$OUT"; mkdir -p "$OUT"
  export CMDIAD_LN_FOLD=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/vit_profile.py vit > "$OUT/run.log" 2>&1
  python3 tools/summarize_profile.py "$OUT"/*/*kernel_trace.csv 13 2>/dev/null | cut -c1-150 | head -16 | tee $O/summary_$mode.md
done
