for i in 1 2 3; do for v in "0 4" "-1 4" "0 8"; do set -- $v; CMDIAD_COPY_PRIO=$1 GPU_MAX_HW_QUEUES=$2 CMDIAD_BENCH_LEGS=h2d_inclusive python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('copy prio $1 hw queues $2: value',d['value'],'ms',d['ms_per_step'],'h2d',d['h2d_inclusive']['value'],'frac',d['roofline']['frac'])"; done; done
