#!/bin/bash
# round 5, GPU call 4: the whole GPU suite with durations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_4
mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -x -q -p no:cacheprovider --durations=40 > $O/t_all.log 2>&1; echo "suite rc=$?" | tee -a $O/rc.log
tail -n 60 $O/t_all.log
