#!/bin/bash
# round 5, GPU call 27: is the drop-in's micro-batching exact?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_27
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_predictor.py -x -q -m gpu -s -k "micro_batching" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
grep -h "micro-batch\|passed\|failed" $O/tests.log | tail -n 6
