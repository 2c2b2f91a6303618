#!/bin/bash
# round 5, GPU call 21: encoder tail with the round-2 tool (same synthetic data as the 2.53-2.60 ms of r2_notes) + ablations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_21
mkdir -p $O
TAIL_ABLATE=1 timeout 600 python tools/tail_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/tail.log
