#!/bin/bash
# round 6, GPU call 1: the runner-up (exact argmin): parity tests, A/B of the distance GEMM against the round-5 kernel, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r6_1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "l2_ or normalize" -s > $O/t_l2.log 2>&1; echo "l2 tests rc=$?" | tee -a $O/rc.log; tail -n 4 $O/t_l2.log | tee -a $O/rc.log
grep "argmin ==" $O/t_l2.log | tee -a $O/rc.log
timeout 600 python tools/l2_runner_ab.py 3 > $O/ab.log 2>&1; echo "ab rc=$?" | tee -a $O/rc.log; cat $O/ab.log | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -x -q -s > $O/t_full.log 2>&1; echo "fullsize+engine rc=$?" | tee -a $O/rc.log; tail -n 4 $O/t_full.log | tee -a $O/rc.log
grep "full size" $O/t_full.log | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_fakeworld.py tests/test_gpu_world2.py -x -q > $O/t_pred.log 2>&1; echo "predictor+fakeworld+world2 rc=$?" | tee -a $O/rc.log; tail -n 4 $O/t_pred.log | tee -a $O/rc.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.log
python - <<'PY' | tee -a $O/rc.log
import json
try:
    d = json.loads(open("gpurun_out/r6_1/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "h2d", d.get("h2d_inclusive"), "mtfi", d.get("mtfi_step", {}).get("value"))
    print({k: (v.get("error") or v.get("skipped")) for k, v in d.items() if isinstance(v, dict) and ("error" in v or "skipped" in v)})
except Exception as e:
    print("bench parse failed", e)
PY
