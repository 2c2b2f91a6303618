#!/bin/bash
# round 5, GPU call 8: cheap switches on the final tree: LayerNorm fold for the ViT, library ranges per query tile on bf16 operands
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_8
mkdir -p $O
timeout 900 bash tools/ab_bench.sh CMDIAD_LN_FOLD "pmae 1" 3 2>&1 | tee -a $O/rc.log
for sp in 10 15 20 25 30 40; do
  echo "splits $sp" | tee -a $O/rc.log
  CMDIAD_L2_SPLITS=$sp timeout 300 python tools/l2_counted.py 2>&1 | grep "counted Q=54401 of 100352" | tail -n 1 | tee -a $O/rc.log
done
for qg in 2 4 8; do
  echo "qgroup $qg" | tee -a $O/rc.log
  CMDIAD_L2_QGROUP=$qg timeout 300 python tools/l2_counted.py 2>&1 | grep "counted Q=54401 of 100352" | tail -n 1 | tee -a $O/rc.log
done
timeout 600 python -m pytest tests/test_gpu_engine.py -k "scoring_exact or public_features" -m gpu -q -p no:cacheprovider > $O/t_eng.log 2>&1; echo "engine rc=$?" | tee -a $O/rc.log
tail -n 3 $O/t_eng.log
