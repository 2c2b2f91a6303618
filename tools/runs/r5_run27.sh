#!/bin/bash
# round 5, GPU call 27: are the drop-in's micro-batching and the engine's batch invariance exact?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_27
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_predictor.py -x -q -m gpu -k "micro_batching" > $O/tests.log 2>&1; echo "predictor rc=$?" | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "batch_invariance or public_features" > $O/engine.log 2>&1; echo "engine rc=$?" | tee -a $O/rc.log
tail -n 12 $O/engine.log
