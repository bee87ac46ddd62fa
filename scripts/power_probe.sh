#!/bin/bash
# samples rocm-smi (power, clocks, temperature) while bench.py runs: is the fit step power-capped?
# usage: scripts/power_probe.sh [bench args...]   -> gpurun_out/power_probe.txt
out=gpurun_out/power_probe.txt
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 150 --warmup 5 "$@" > gpurun_out/power_probe_bench.json 2>/dev/null &
bp=$!
sleep 4
for i in $(seq 1 8); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done > $out
wait $bp
echo "idle:" >> $out
sleep 2
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | tr '\n' ';' >> $out
echo >> $out
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" >> $out
cat $out
python -c "
import json; d=json.load(open('gpurun_out/power_probe_bench.json')); print('bench', d['value'], d['ms_per_step'])"
