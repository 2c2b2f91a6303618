#!/bin/bash
# Profiles of the bench command for one round (run ON the MI355X box from the repo root):
#   tools/profile_round.sh r1      -> gpurun_out/prof_r1/{trace,pmc/<pass>}
# then, back in the dev container:
#   python tools/summarize_profile.py gpurun_out/prof_r1/trace/*/*kernel_trace.csv 12 > profiles/r1_summary.md
#   python tools/pmc_summary.py gpurun_out/prof_r1/pmc profiles/r1_pmc.json > profiles/r1_pmc.md
#   python tools/standalone_summary.py gpurun_out/prof_r1/standalone gpurun_out/prof_r1/standalone_pmc,gpurun_out/prof_r1/standalone_l2_pmc/fetch,gpurun_out/prof_r1/standalone_l2_pmc/write gpurun_out/prof_r1/standalone_work.json > profiles/r1_standalone.md
#   python tools/l2_standalone_pmc.py gpurun_out/prof_r1 profiles/r1   (-> profiles/r1_l2_standalone.json + the meta file bench.py reads)
# Counter passes are separate runs (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2) and never combined with
# sys/hip/hsa tracing.  The program sits directly after `--` (no env/bash hop after the profiler preloads).
R=${1:-r1}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" | cut -c1-400
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" \
            "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    name=${pass%%:*}
    ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/pmc/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$name.log" 2>&1
    echo "pass $name rc=$?"
done
du -sh "$OUT"
# stand-alone kernels: one kernel trace + FETCH / WRITE passes of the two HBM-bound kernels
export STANDALONE_WORK_DIR="$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/standalone" -- python3 tools/standalone_kernels.py > "$OUT/standalone.log" 2>&1
echo "standalone trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/standalone_pmc/fetch" -- python3 tools/standalone_kernels.py hbm > "$OUT/standalone_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/standalone_pmc/write" -- python3 tools/standalone_kernels.py hbm > "$OUT/standalone_write.log" 2>&1
echo "standalone pmc rc=$?"
# the dominant kernel (bench.py's `roofline`) ALONE at its four shapes: the same counter passes as for the bench run, so that
# `roofline.traffic` and the issue / stall split are of the regime `roofline.frac` is quoted in
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" \
            "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    name=${pass%%:*}
    ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/standalone_l2_pmc/$name" -- python3 tools/standalone_kernels.py l2 > "$OUT/standalone_l2_$name.log" 2>&1
    echo "standalone l2 pass $name rc=$?"
done
du -sh "$OUT"
