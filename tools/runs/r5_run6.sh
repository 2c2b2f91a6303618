#!/bin/bash
# round 5, GPU call 6: fixed tests with diagnostics; bf16 against fp16 search operands (kernel alone, and the bench line)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_heads.py::test_dumped_pairs_round_trip_into_the_head_trainers tests/test_gpu_nets.py -k "heavy or stage_by_stage or dumped" -m gpu -q -p no:cacheprovider -s > $O/t_fix.log 2>&1; echo "fix rc=$?" | tee -a $O/rc.log
grep -h "heavy-tailed" $O/t_fix.log | tee -a $O/rc.log
tail -n 5 $O/t_fix.log
for i in 1 2 3; do
  CMDIAD_SEARCH_DTYPE=fp16 timeout 300 python tools/l2_counted.py > $O/dt_fp16_$i.log 2>&1
  CMDIAD_SEARCH_DTYPE=bf16 timeout 300 python tools/l2_counted.py > $O/dt_bf16_$i.log 2>&1
done
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/dt_fp16_*.log | sed 's/^/fp16: /' | tee -a $O/rc.log
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/dt_bf16_*.log | sed 's/^/bf16: /' | tee -a $O/rc.log
timeout 900 bash tools/ab_bench.sh CMDIAD_SEARCH_DTYPE "fp16 bf16" 3 2>&1 | tee -a $O/rc.log
CMDIAD_SEARCH_DTYPE=bf16 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_predictor.py tests/test_gpu_fakeworld.py -k "l2_min or b32 or mtfi_batch or heavy or fake" -m gpu -q -p no:cacheprovider -s > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?" | tee -a $O/rc.log
grep -h "image score\|AUROC\|heavy-tailed" $O/t_bf16.log | tee -a $O/rc.log
tail -n 8 $O/t_bf16.log
