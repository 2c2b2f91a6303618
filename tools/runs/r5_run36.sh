#!/bin/bash
# round 5, GPU call 36: the whole GPU suite after the shared-stream change + the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_36
mkdir -p $O
timeout 1700 python -m pytest tests/ -m gpu -q -p no:cacheprovider --durations=12 > $O/t_all.log 2>&1; echo "suite rc=$?" | tee -a $O/rc.log
tail -n 22 $O/t_all.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/rc.log
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], [k for k,v in d.items() if isinstance(v,dict) and 'error' in v])"
