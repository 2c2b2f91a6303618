#!/bin/bash
# round 5, GPU call 25: randomised exact invariants of the batched pipeline
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_25
mkdir -p $O
timeout 900 python tools/fuzz_pipeline.py 420 5 > $O/fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $O/rc.log
tail -n 12 $O/fuzz.log | cut -c1-400
