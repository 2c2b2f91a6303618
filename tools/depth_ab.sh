for i in 1 2 3; do for d in 2 3; do CMDIAD_BENCH_DEPTH=$d CMDIAD_BENCH_LEGS=h2d_inclusive python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $d: value',d['value'],'ms',d['ms_per_step'],'h2d',d['h2d_inclusive']['value'],'frac',d['roofline']['frac'])"; done; done
