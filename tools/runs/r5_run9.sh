#!/bin/bash
# round 5, GPU call 9: randomised sweep of the new pieces (distance-GEMM key identity, pair scan), then every case
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_9
mkdir -p $O
FUZZ_ONLY=l2_identity,reweight_pair timeout 600 python tools/fuzz_gpu.py 240 11 > $O/fuzz_new.log 2>&1; echo "fuzz new rc=$?" | tee -a $O/rc.log
tail -n 3 $O/fuzz_new.log | tee -a $O/rc.log
timeout 600 python tools/fuzz_gpu.py 240 12 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?" | tee -a $O/rc.log
tail -n 3 $O/fuzz_all.log | tee -a $O/rc.log
