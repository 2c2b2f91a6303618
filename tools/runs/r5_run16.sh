#!/bin/bash
# round 5, GPU call 16: attention with s_setprio around the MFMA groups
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_16
mkdir -p $O
for i in 1 2; do for n in 0 p1 p2; do
  CMDIAD_HIP_LIB=$PWD/tools/_ab/libatt_$n.so python tools/attbench.py 2>&1 | grep "0.18" | tee -a $O/abl.log
done; done
