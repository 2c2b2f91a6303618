#!/bin/bash
# round 5, GPU call 29: soak of the pipelined step (both workloads)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_29
mkdir -p $O
timeout 600 python tools/soak.py 5000 dino_pointmae 2>&1 | grep -v amdgpu.ids | tee $O/soak_a.log | tail -n 4
timeout 600 python tools/soak.py 3000 mtfi 2>&1 | grep -v amdgpu.ids | tee $O/soak_b.log | tail -n 3
