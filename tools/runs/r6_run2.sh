#!/bin/bash
# round 6, GPU call 2: the whole GPU suite with per-test durations (what to trim), then the round's profile passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r6_2
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/rc.log
tail -n 75 $O/gpu_suite.log | tee -a $O/rc.log
