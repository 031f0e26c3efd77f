#!/bin/bash
# Samples sclk / power while the recogniser runs flat out (is the f32 MFMA peak reachable at the clocks this load gets?)
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --workload embed --steps 300 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/clock_bench.json 2> $R/gpurun_out/clock_bench.err &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -6
  echo ---
  sleep 1
done
wait $PID
tail -c 300 $R/gpurun_out/clock_bench.json
echo
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
