#!/bin/bash
# round 5, GPU call 24: DepthFeatures / interpolate_points (drop-in surface additions)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_24
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "depth_features or other_method_classes or public_features" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/rc.log
tail -n 12 $O/tests.log
