#!/bin/bash
# round 5, GPU call 34: accumulator zeroing with v_mov_b64 in the 128 x 128 kernels: ViT / Point-MAE forward and bench A/B, kernel tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_34
mkdir -p $O
OLD=$PWD/tools/_ab/libzero32.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or qkv or conv" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 3 $O/tests.log
for i in 1 2 3; do
  for w in vit pmae; do
    CMDIAD_HIP_LIB=$OLD python tools/vit_profile.py $w 2>&1 | grep "per forward" | sed 's/^/mov_b32  /' | tee -a $O/fw.log
    python tools/vit_profile.py $w 2>&1 | grep "per forward" | sed 's/^/mov_b64  /' | tee -a $O/fw.log
  done
done
for i in 1 2 3; do
  CMDIAD_HIP_LIB=$OLD python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mov_b32', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mov_b64', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
