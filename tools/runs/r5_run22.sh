#!/bin/bash
# round 5, GPU call 22: encoder tail after the final-reduction fix (16 instead of 32 values per chunk through the DPP maxima)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_22
mkdir -p $O
timeout 600 python tools/tail_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/tail.log
timeout 900 python -m pytest tests/ -x -q -m gpu -k "encoder or tail or pointmae or pmae" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 4 $O/tests.log
FUZZ_ONLY=encoder timeout 300 python tools/fuzz_gpu.py 90 31 2>&1 | tail -n 1 | tee -a $O/rc.log
