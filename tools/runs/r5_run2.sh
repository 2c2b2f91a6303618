#!/bin/bash
# round 5, GPU call 2: 160-row tiles for the N = 768 residual products; the new bench legs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r5_2
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "tile_height or gemm_epilogues or l2_min_key" -m gpu -x -q -p no:cacheprovider > $O/t_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/rc.log
timeout 600 python tools/res_bm_ab.py > $O/res_bm_ab.log 2>&1; echo "res_bm_ab rc=$?" | tee -a $O/rc.log
cat $O/res_bm_ab.log | tee -a $O/rc.log
timeout 900 python -m pytest tests/test_gpu_nets.py -m gpu -x -q -p no:cacheprovider > $O/t_nets.log 2>&1; echo "nets rc=$?" | tee -a $O/rc.log
timeout 900 bash tools/ab_bench.sh CMDIAD_GEMM_RES_BM "128 160" 3 2>&1 | tee -a $O/rc.log
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?" | tee -a $O/rc.log
tail -c 600 $O/bench_full.err
for f in t_kernels t_nets; do tail -n 3 $O/$f.log; done
