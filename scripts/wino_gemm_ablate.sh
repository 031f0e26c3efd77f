#!/bin/bash
# Ablation timing of wino_gemm_kernel<64, 3> (winograd.hip): a diagnostic build (-DFACEHIP_WINO_ABL) installed as a SIDE copy of the
# library (FACEHIP_LIB), then the recogniser's layer table per switch combination (bits: 1 = no loads in the K loop, 2 = no LDS reads,
# 4 = no barriers in the K loop, 8 = no stores, 16 = stores straight from the accumulator registers).
# FACEHIP_WINO_BN128=0: the ablated instantiations exist for the 128 x 64 tile only.  Results of an ablated run are wrong by construction; only the kernel times are read.
set -e
cd "$(dirname "$0")/.."
B=build/facehip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DFACEHIP_WINO_ABL -c facerecognizeonnx_amd/csrc/winograd.hip -o $B/winograd_abl.o
OBJS=$(ls $B/*.o | grep -v "/winograd.o" | grep -v "_prof.o" | grep -v "_abl.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libfacehip_wabl.so $OBJS $B/winograd_abl.o -lz
for abl in ${@:-0 1 2 4 8 5 3 6 7 15}; do
  echo "== ablation=$abl"
  FACEHIP_WINO_BN128=0 FACEHIP_WINO_ABL=$abl FACEHIP_LIB=/tmp/libfacehip_wabl.so PYTHONPATH=. python scripts/layer_times.py rec 128 2>&1 | grep -E "cfg7  (20|21|49) " | cut -c1-70
done
