#!/bin/bash
# round 5, GPU call 15: attention timing-only ablations + occupancy; the new attention tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_15
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nets.py -x -q -m gpu -s -k "attention or stage_by_stage" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
grep -h "attention model\|passed\|failed" $O/tests.log | tail -n 8
for n in 0 1 2 3 4 5 6; do
  CMDIAD_HIP_LIB=$PWD/tools/_ab/libatt_$n.so python tools/attbench.py 2>&1 | grep "0.18" | tee -a $O/abl.log
done
for occ in 2 3 4; do CMDIAD_ATT_OCC=$occ python tools/attbench.py 2>&1 | grep "0.18" | sed "s/^/occ $occ /" | tee -a $O/abl.log; done
