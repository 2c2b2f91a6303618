#!/bin/bash
# round 5, GPU call 10: two REAL ranks on one GPU (gloo transport) through the row-sharded paths
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_10
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_world2.py tests/test_gpu_fakeworld.py -k "world2 or sharded_fp32" -m gpu -x -q -p no:cacheprovider > $O/t.log 2>&1; echo "world2 rc=$?" | tee -a $O/rc.log
tail -n 40 $O/t.log
