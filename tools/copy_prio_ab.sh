# A/B on the bench line: host batches staged two submits ahead on the post stream (CMDIAD_BENCH_STAGE_AHEAD=1) against the copy
# stream at submit time (0); earlier alternations of this script (copy-stream priority, GPU_MAX_HW_QUEUES) are in profiles/r6_notes.md
for i in 1 2 3; do for v in 1 0; do CMDIAD_BENCH_STAGE_AHEAD=$v CMDIAD_BENCH_LEGS=h2d_inclusive python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage_ahead $v: value',d['value'],'ms',d['ms_per_step'],'h2d',d['h2d_inclusive']['value'],'frac',d['roofline']['frac'])"; done; done
