#!/bin/bash
# sample the GPU clocks / power while a kernel loop runs (box only)
"$@" > gpurun_out/clock_probe_run.log 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8
  echo ---
  sleep 1
done
wait $pid
tail -2 gpurun_out/clock_probe_run.log
