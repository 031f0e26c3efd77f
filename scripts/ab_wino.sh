#!/bin/bash
# A/B of the Winograd GEMM forms on IResNet-50 (tuning aid): per-layer times with the persistent kernel on / off.
set -e
mkdir -p gpurun_out
for st in 0 1 2 4 8; do
  FACEHIP_WINO_STAGGER=$st FACEHIP_WINO_PERSIST=1 python scripts/layer_times.py rec ${1:-128} > gpurun_out/lt_persist1_$st.txt 2>&1
  echo "stagger $st: $(tail -1 gpurun_out/lt_persist1_$st.txt) $(grep 'cfg7 30' gpurun_out/lt_persist1_$st.txt | cut -c1-30)"
done
FACEHIP_WINO_PERSIST=0 python scripts/layer_times.py rec ${1:-128} > gpurun_out/lt_persist0.txt 2>&1
echo "plain: $(tail -1 gpurun_out/lt_persist0.txt) $(grep 'cfg7 30' gpurun_out/lt_persist0.txt | cut -c1-30)"
