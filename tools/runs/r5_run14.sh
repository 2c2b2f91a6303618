#!/bin/bash
# round 5, GPU call 14: attention with a lazily moved softmax reference: parity, kernel A/B, bench A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_14
mkdir -p $O
OLD=$PWD/tools/_ab/libcmdiad_hip_attold.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q -m gpu -k "attention or vit or pmae or pointmae or forward" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 5 $O/tests.log
for i in 1 2 3; do
  CMDIAD_HIP_LIB=$OLD python tools/attbench.py 2>&1 | grep attention | tee -a $O/att.log
  python tools/attbench.py 2>&1 | grep attention | tee -a $O/att.log
done
for i in 1 2 3; do
  CMDIAD_HIP_LIB=$OLD python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
