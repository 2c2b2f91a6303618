# The bench line N times (separate processes, --no-extras): the run-to-run distribution of `value` on one box
N=${1:-12}
for i in $(seq $N); do python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i: value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'in_pipe',d['roofline']['frac_in_pipeline'])"; done
