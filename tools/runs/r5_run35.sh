#!/bin/bash
# round 5, GPU call 35: V tiles of the qkv projection through the LDS scratch as 128-byte token rows: parity and A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_35
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q -m gpu -k "qkv or stage_by_stage or block or forward or attention" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 4 $O/tests.log
for i in 1 2 3; do
  CMDIAD_QKV_VROWS=0 python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/direct  /' | tee -a $O/fw.log
  python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/v rows  /' | tee -a $O/fw.log
done
for i in 1 2 3; do
  CMDIAD_QKV_VROWS=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v rows', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
