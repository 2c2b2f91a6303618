export TMPDIR=/tmp
for t in 4 5; do
OUT=$PWD/gpurun_out/pmc_l2_$t; rm -rf $OUT; mkdir -p $OUT
L2_VARIANTS=$t L2_Q=25088 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 tools/l2_ab.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*counter_collection.csv")[0]
agg=collections.defaultdict(float); n=0
for r in csv.DictReader(open(f)):
    if "l2_min_pp" in r["Kernel_Name"]:
        agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print("tile $t", {k: v for k,v in agg.items()})
PY
done
