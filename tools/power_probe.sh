#!/bin/bash
# sample package power / clocks while bench.py loops (diagnostic: is the forward power-capped?)
# usage: bash tools/power_probe.sh [extra bench.py args]   -> gpurun_out/power.log
python bench.py --no-cpu-baseline --steps ${STEPS:-2500} "$@" > gpurun_out/bench_long.log 2>&1 &
BP=$!
sleep 30
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|junction" >> gpurun_out/power.log
  echo "---" >> gpurun_out/power.log
  sleep 2
done
wait $BP
rocm-smi -M 2>&1 | grep -i "power" >> gpurun_out/power.log 2>&1
tail -1 gpurun_out/bench_long.log | cut -c1-160
