#!/bin/bash
# Samples the GPU's power / clocks / temperature (rocm-smi, read-only) while bench.py runs (ON the MI355X box):  tools/power_probe.sh
mkdir -p gpurun_out/power
python bench.py --no-extras --no-cpu-baseline --steps 1500 --warmup 4 > gpurun_out/power/bench.json 2> gpurun_out/power/bench.err &
BP=$!
sleep 22
for i in $(seq 30); do
  rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)|GPU use" | tr -s " " | tr "\n" ";"
  echo
  sleep 1
done
wait $BP
cut -c1-160 gpurun_out/power/bench.json
echo "--- idle:"
sleep 3
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr -s " " | tr "\n" ";"; echo
