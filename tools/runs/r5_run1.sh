#!/bin/bash
# round 5, GPU call 1: the new running minimum of the distance GEMM -- parity, A/B against the round-4 build, stand-alone profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -k "l2_min" -m gpu -x -q -p no:cacheprovider > $O/t_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/rc.log
CMDIAD_TEST_AB=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -k "l2_min" -m gpu -x -q -p no:cacheprovider > $O/t_kernels_ab.log 2>&1; echo "kernels_ab rc=$?" | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_fakeworld.py tests/test_gpu_dedup.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > $O/t_world.log 2>&1; echo "world rc=$?" | tee -a $O/rc.log
for i in 1 2 3; do
  CMDIAD_HIP_LIB=$PWD/tools/_ab/libcmdiad_hip_r4.so timeout 300 python tools/l2_counted.py > $O/ab_old_$i.log 2>&1
  timeout 300 python tools/l2_counted.py > $O/ab_new_$i.log 2>&1
done
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/ab_old_*.log | sed 's/^/old: /' | tee -a $O/rc.log
grep -h "counted Q=54401 of 100352\|plain   Q=100352" $O/ab_new_*.log | sed 's/^/new: /' | tee -a $O/rc.log
export STANDALONE_WORK_DIR=$O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/l2_trace -- python3 $GRAFT_REPO_ROOT/tools/standalone_kernels.py l2 > $GRAFT_REPO_ROOT/$O/l2_trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/l2_pmc/fetch -- python3 $GRAFT_REPO_ROOT/tools/standalone_kernels.py l2 > $GRAFT_REPO_ROOT/$O/l2_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/l2_pmc/write -- python3 $GRAFT_REPO_ROOT/tools/standalone_kernels.py l2 > $GRAFT_REPO_ROOT/$O/l2_write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
tail -3 $O/t_kernels.log $O/t_kernels_ab.log $O/t_world.log
du -sh $O
