#!/bin/bash
# ablation of dwpw_reg_kernel (tuning aid): per-layer times of the detector's stride-1 blocks with parts of the kernel switched off
for d in 0 1 2 3 4 8 15; do
  echo "== FACEHIP_DWPW_DBG=$d"
  FACEHIP_DWPW_DBG=$d PYTHONPATH=. python scripts/layer_times.py det 128 2>&1 | grep -E "^ +[0-9.]+ us.*(  3 DW|  5 DW|  33 DW|  34 DW)" | cut -c1-100
done
