#!/bin/bash
# round 5, GPU call 13: bench.py --gpus 2 rehearsed on one device (both bank modes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_13
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_world2.py -x -q -m gpu -s --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 40 $O/tests.log
