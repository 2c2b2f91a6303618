#!/bin/bash
# round 5, GPU call 5: the rest of the GPU suite with durations, then the round's profiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_5
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_heads.py tests/test_gpu_kernels.py tests/test_gpu_lnfold.py tests/test_gpu_nets.py tests/test_gpu_ocsvm.py tests/test_gpu_predictor.py tests/test_gpu_train.py tests/test_gpu_variants.py -m gpu -q -p no:cacheprovider --durations=25 > $O/t_rest.log 2>&1; echo "rest rc=$?" | tee -a $O/rc.log
tail -n 45 $O/t_rest.log
timeout 2400 bash tools/profile_round.sh r5 2>&1 | tee -a $O/rc.log
timeout 600 python bench.py > gpurun_out/prof_r5/bench_default.json 2> gpurun_out/prof_r5/bench_default.err; echo "bench rc=$?" | tee -a $O/rc.log
