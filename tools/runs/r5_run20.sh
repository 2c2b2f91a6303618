#!/bin/bash
# round 5, GPU call 20: randomised sweep on the final tree (attention with scaled / climbing scores first, then every case)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_20
mkdir -p $O
FUZZ_ONLY=attention timeout 400 python tools/fuzz_gpu.py 180 21 > $O/fuzz_att.log 2>&1; echo "fuzz attention rc=$?" | tee -a $O/rc.log
tail -n 3 $O/fuzz_att.log | tee -a $O/rc.log
timeout 500 python tools/fuzz_gpu.py 240 22 > $O/fuzz_all.log 2>&1; echo "fuzz all rc=$?" | tee -a $O/rc.log
tail -n 3 $O/fuzz_all.log | tee -a $O/rc.log
