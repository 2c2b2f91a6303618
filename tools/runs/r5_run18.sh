#!/bin/bash
# round 5, GPU call 18: the whole GPU suite (durations), the round's profiles with the final defaults, the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_18
mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q -p no:cacheprovider --durations=15 > $O/t_all.log 2>&1; echo "suite rc=$?" | tee -a $O/rc.log
tail -n 30 $O/t_all.log
rm -rf gpurun_out/prof_r5
timeout 2400 bash tools/profile_round.sh r5 2>&1 | tee -a $O/rc.log
timeout 900 python bench.py > gpurun_out/prof_r5/bench_default.json 2> gpurun_out/prof_r5/bench_default.err; echo "bench rc=$?" | tee -a $O/rc.log
