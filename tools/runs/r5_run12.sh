#!/bin/bash
# round 5, GPU call 12: two real ranks after the redo fix; the fake-world and predictor tests again
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world2.py tests/test_gpu_fakeworld.py tests/test_gpu_predictor.py -x -q -m gpu --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 25 $O/tests.log
