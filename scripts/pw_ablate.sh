#!/bin/bash
# Ablation timing of conv_pw_kernel (conv_mfma.hip): a diagnostic build (-DFACEHIP_PW_ABL) installed as a SIDE copy of the library
# (FACEHIP_LIB), then the detector's layer table per switch combination (bits: 1 = no loads in the K loop, 2 = no LDS reads,
# 4 = no barriers in the K loop, 8 = no epilogue).  Results of an ablated run are wrong by construction; only the kernel times are read.
set -e
cd "$(dirname "$0")/.."
B=build/facehip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DFACEHIP_PW_ABL -c facerecognizeonnx_amd/csrc/conv_mfma.hip -o $B/conv_mfma_abl.o
OBJS=$(ls $B/*.o | grep -v "/conv_mfma.o" | grep -v "_prof.o" | grep -v "_abl.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libfacehip_pwabl.so $OBJS $B/conv_mfma_abl.o -lz
for abl in ${@:-0 1 2 4 8 7 15 9}; do
  echo "== ablation=$abl"
  FACEHIP_PW_ABL=$abl FACEHIP_LIB=/tmp/libfacehip_pwabl.so PYTHONPATH=. python scripts/layer_times.py det 128 2>&1 | grep -E "cfg11  (10|14|16) " | cut -c1-70
done
echo "== phase stamps (ablation=${PHASE_ABL:-0})"
FACEHIP_PW_ABL=${PHASE_ABL:-0} FACEHIP_LIB=/tmp/libfacehip_pwabl.so PYTHONPATH=. python scripts/pw_phases.py 2>&1 | grep -v amdgpu.ids
