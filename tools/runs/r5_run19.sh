#!/bin/bash
# round 5, GPU call 19: two real ranks again (sharded-fp32 pipeline + its replicated-queries guard)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_19
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world2.py -x -q -m gpu -k "rehearsed" -s > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 15 $O/tests.log
