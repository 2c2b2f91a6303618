#!/bin/bash
# round 5, GPU call 17: xyz_patch_fused with 64 / 128 / 256 threads per patch; kernel and bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_17
mkdir -p $O
for nt in 256 128 64; do CMDIAD_XYZ_PATCH_THREADS=$nt python tools/xyz_patch_time.py 2>&1 | grep threads | tee -a $O/xyz.log; done
for i in 1 2; do for nt in 256 128 64; do
  CMDIAD_XYZ_PATCH_THREADS=$nt python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $nt', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done; done
