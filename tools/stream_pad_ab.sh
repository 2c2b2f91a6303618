# Which pool stream (hence which hardware queue) the predictor's three streams land on: CMDIAD_STREAM_PAD sweeps, bench line per case
for rep in 1 2; do
for pad in "" "predictor.side=1" "predictor.side=2" "predictor.side=3" "predictor.post=1" "predictor.post=2" "predictor.post=3" "predictor.copy=1" "predictor.copy=2" "predictor.copy=3" "predictor.side=1,predictor.post=1" "predictor.side=2,predictor.post=2"; do
CMDIAD_STREAM_PAD="$pad" CMDIAD_BENCH_LEGS=h2d_inclusive python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad [$pad]: value',d['value'],'ms',d['ms_per_step'],'h2d',d['h2d_inclusive']['value'],'frac',d['roofline']['frac'],'in_pipe',d['roofline']['frac_in_pipeline'])"
done; done
