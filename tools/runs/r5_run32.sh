#!/bin/bash
# round 5, GPU call 32: LayerNorm with 16-byte accesses (C % 256 == 0): parity, ViT forward A/B, bench A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_32
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q -m gpu -k "layernorm or vit or stage_by_stage or forward" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 4 $O/tests.log
for i in 1 2 3; do
  CMDIAD_LN_WIDE=0 python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/float2  /' | tee -a $O/vit.log
  python tools/vit_profile.py vit 2>&1 | grep "per forward" | sed 's/^/float4  /' | tee -a $O/vit.log
done
for i in 1 2 3; do
  CMDIAD_LN_WIDE=0 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('float2', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('float4', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
