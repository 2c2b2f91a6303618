#!/bin/bash
# Distance GEMM: library ranges per query tile (CMDIAD_L2_SPLITS, read once per process) against time at the compacted and the full row count.
# L2_NB (default 76544 = the bagel xyz library padded to whole 256-row tiles, engine.Bank) sets the library rows.
for sp in ${SPLITS:-12 16 18 20 22 24 28 32}; do
  CMDIAD_L2_SPLITS=$sp timeout 200 python tools/l2_counted.py 2>&1 | grep -E "counted Q=54401 of 100352|plain   Q=100352" | tail -2 | sed "s/^/splits=$sp /"
done
