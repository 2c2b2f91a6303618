#!/bin/bash
# ONE parametrised runner for a `gpurun` call (replaces the per-call scripts tools/runs/rN_runM.sh of rounds 5-6):
#   gpurun --timeout 1800 -- 'bash tools/gpu_session.sh r6_7 "pytest:l2_ or normalize" "py:tools/l2_runner_ab.py 3" bench "ab:CMDIAD_SEARCH_DTYPE:bf16 fp16:2"'
# First argument: the output directory under gpurun_out/ (merged back into the dev container); every further argument is a step,
# run in order, each with its own log file and a line (+ the log's tail) in <out>/rc.log:
#   suite[:extra pytest args]        python -m pytest tests -m gpu -q --durations=25
#   pytest:<-k expression>[:files]   python -m pytest <files or tests> -m gpu -q -s -k <expression>
#   bench[:args]                     python bench.py <args>  (default: --no-cpu-baseline) + a one-line summary of the JSON line
#   ab:<VAR>:<values>[:passes]       tools/ab_bench.sh (one environment switch on the bench line, candidates interleaved)
#   profile:<rN>                     tools/profile_round.sh rN (kernel trace + counter passes of the bench and of the stand-alone kernels)
#   py:<script args>                 python <script args>
#   sh:<command>                     bash -c <command>
# Steps never stop the session; per-step wall-clock limit STEP_TIMEOUT (default 1500 s).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p "$O"
LIM=${STEP_TIMEOUT:-1500}
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; rest=""; [ "$kind" != "$step" ] && rest=${step#*:}
  log="$O/${n}_${kind}.log"
  case "$kind" in
    suite)   timeout $LIM python -m pytest tests -m gpu -q --durations=25 $rest > "$log" 2>&1 ;;
    pytest)  expr=${rest%%:*}; files=tests; [ "$expr" != "$rest" ] && files=${rest#*:}
             timeout $LIM python -m pytest $files -m gpu -q -s -k "$expr" > "$log" 2>&1 ;;
    bench)   timeout $LIM python bench.py ${rest:---no-cpu-baseline} > "$O/${n}_bench.json" 2> "$log"; rc=$?
             python - "$O/${n}_bench.json" >> "$log" 2>&1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
legs = {k: v.get("value", v.get("ms_per_step")) for k, v in d.items() if isinstance(v, dict) and ("value" in v or "ms_per_step" in v)}
print("BENCH value", d.get("value"), "ms/step", d.get("ms_per_step"), "frac", r.get("frac"), "launch_ms", r.get("launch_ms"), "in_pipeline", r.get("frac_in_pipeline"))
print("BENCH legs", legs)
print("BENCH errors", {k: (v.get("error") or v.get("skipped")) for k, v in d.items() if isinstance(v, dict) and ("error" in v or "skipped" in v)})
PY
             (exit $rc) ;;
    ab)      IFS=: read -r var vals passes <<< "$rest"; bash tools/ab_bench.sh "$var" "$vals" ${passes:-2} > "$log" 2>&1 ;;
    profile) bash tools/profile_round.sh "$rest" > "$log" 2>&1; find gpurun_out/prof_$rest -name "*.db" -delete 2>/dev/null ;;
    py)      timeout $LIM python $rest > "$log" 2>&1 ;;
    sh)      timeout $LIM bash -c "$rest" > "$log" 2>&1 ;;
    *)       echo "unknown step $step" > "$log"; false ;;
  esac
  echo "step $n [$step] rc=$?" | tee -a "$O/rc.log"
  tail -n ${TAIL_LINES:-12} "$log" | cut -c1-400 | tee -a "$O/rc.log"
done
