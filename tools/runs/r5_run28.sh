#!/bin/bash
# round 5, GPU call 28: randomised exact invariant of the drop-in method classes (micro-batch size)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_28
mkdir -p $O
timeout 900 python tools/fuzz_dropin.py 360 3 > $O/fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $O/rc.log
grep -v "^frame\|^$\|Warning\|warnings.warn" $O/fuzz.log | tail -n 15 | cut -c1-400
