# A/B on the bench line: the next batch's copy staged ahead of the step (CMDIAD_BENCH_STAGE_NEXT), hardware queue count
for i in 1 2 3; do for v in "1 4" "0 4" "1 2" "1 3" "1 6"; do set -- $v; CMDIAD_BENCH_STAGE_NEXT=$1 GPU_MAX_HW_QUEUES=$2 CMDIAD_BENCH_LEGS=h2d_inclusive python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage_next $1 hw queues $2: value',d['value'],'ms',d['ms_per_step'],'h2d',d['h2d_inclusive']['value'],'frac',d['roofline']['frac'])"; done; done
