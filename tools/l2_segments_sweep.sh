mkdir -p gpurun_out/r4b
for qg in 4 8; do
for sp in 1 2 3 4 5 6 8 9 10 12 16 20; do
  echo "== qgroup $qg splits $sp" >> gpurun_out/r4b/sweep.log
  CMDIAD_L2_QGROUP=$qg CMDIAD_L2_SEG_SPLITS=$sp python tools/l2_segments.py bagel 1,2,4,8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['cls'],d['world'],d['shard_tiles'],d['gemm_ms_slowest_rank'],d['gemm_frac_of_peak'])
" >> gpurun_out/r4b/sweep.log
done
done
cat gpurun_out/r4b/sweep.log
