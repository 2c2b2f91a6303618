#!/bin/bash
# A/B of one environment switch on the bench line (run ON the MI355X box):  tools/ab_bench.sh CMDIAD_GEMM_RES_WIDE "0 1" [repeats]
VAR=$1; VALS=$2; REP=${3:-2}
mkdir -p gpurun_out/ab
for i in $(seq $REP); do
  n=0
  for v in $VALS; do
    n=$((n+1))
    env $VAR=$v python bench.py --no-extras --no-cpu-baseline > gpurun_out/ab/out_${n}_$i.json 2> gpurun_out/ab/err_${n}_$i.log
    python - "$VAR=$v" gpurun_out/ab/out_${n}_$i.json gpurun_out/ab/err_${n}_$i.log <<'PY'
import json, sys
tag, out, err = sys.argv[1:4]
try:
    d = json.loads(open(out).read().strip().splitlines()[-1])
    print(tag, "images/s", d["value"], "ms/step", d["ms_per_step"], "l2 frac", d["roofline"]["frac"], flush=True)
except Exception as e:
    print(tag, "FAILED", e, open(err).read()[-300:].replace("\n", " | "), flush=True)
PY
  done
done
