#!/bin/bash
# round 5, GPU call 3: quartered running minimum (A/B against the un-quartered build), pair scan, heavy-tailed parity, pair datasets
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "l2_min or reweight" -m gpu -x -q -p no:cacheprovider > $O/t_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/rc.log
CMDIAD_TEST_AB=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -k "l2_min" -m gpu -x -q -p no:cacheprovider > $O/t_kernels_ab.log 2>&1; echo "kernels_ab rc=$?" | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_fakeworld.py tests/test_gpu_dedup.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > $O/t_world.log 2>&1; echo "world rc=$?" | tee -a $O/rc.log
for i in 1 2 3; do
  CMDIAD_HIP_LIB=$PWD/tools/_ab/libcmdiad_hip_r5a.so timeout 300 python tools/l2_counted.py > $O/ab_old_$i.log 2>&1
  timeout 300 python tools/l2_counted.py > $O/ab_new_$i.log 2>&1
done
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/ab_old_*.log | sed 's/^/r5a (one epilogue): /' | tee -a $O/rc.log
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/ab_new_*.log | sed 's/^/quartered: /' | tee -a $O/rc.log
L2_VARIANTS=2,5 L2_STRESS=40 L2_Q=54401 CMDIAD_HIP_LIB=$PWD/cmdiad_amd/libcmdiad_hip_ab.so timeout 600 python tools/l2_ab.py > $O/l2_stress.log 2>&1; tail -n 4 $O/l2_stress.log | tee -a $O/rc.log
timeout 1500 python -m pytest tests/test_gpu_nets.py tests/test_gpu_heads.py -m gpu -x -q -p no:cacheprovider -s > $O/t_nets.log 2>&1; echo "nets+heads rc=$?" | tee -a $O/rc.log
grep -h "heavy-tailed" $O/t_nets.log | tee -a $O/rc.log
timeout 1500 python -m pytest tests/test_gpu_predictor.py -k "heavy or b32 or mtfi_batch" -m gpu -x -q -p no:cacheprovider -s > $O/t_pred.log 2>&1; echo "pred rc=$?" | tee -a $O/rc.log
grep -h "heavy-tailed\|image score" $O/t_pred.log | tee -a $O/rc.log
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?" | tee -a $O/rc.log
python - <<'PY' | tee -a gpurun_out/r5_3/rc.log
import json
d=json.loads(open('gpurun_out/r5_3/bench1.json').read().strip().splitlines()[-1])
print("bench", d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_in_pipeline'], d['roofline']['launch_ms'])
PY
for f in t_kernels t_kernels_ab t_world t_nets t_pred; do tail -n 3 $O/$f.log; done
