#!/bin/bash
# round 6, GPU call 3: the rest of the GPU suite after the first failure of call 2, then the round's profile passes (tools/profile_round.sh r6)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r6_3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/rc.log
tail -n 40 $O/gpu_suite.log | tee -a $O/rc.log
bash tools/profile_round.sh r6 2>&1 | tail -n 30 | tee -a $O/rc.log
# keep the merged output small: the summaries need the csv files only
find gpurun_out/prof_r6 -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/prof_r6 | tee -a $O/rc.log
