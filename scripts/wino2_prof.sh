#!/bin/bash
# Ablation timing of wino2_kernel (conv_wino2.hip): a diagnostic build (-DFACEHIP_W2_PROF) installed as a SIDE copy of the library
# (FACEHIP_LIB), then scripts/wino2_prof.py per switch combination.  Results of an ablated run are wrong by construction.
set -e
cd "$(dirname "$0")/.."
B=build/facehip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DFACEHIP_W2_PROF -c facerecognizeonnx_amd/csrc/conv_wino2.hip -o $B/conv_wino2_prof.o
OBJS=$(ls $B/*.o | grep -v "conv_wino2.o" | grep -v "_prof.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libfacehip_w2prof.so $OBJS $B/conv_wino2_prof.o -lz
for shape in "${@:-56 56 64 64}"; do
  for abl in 0 2 4 8 16 30; do
    echo "== $shape  ablation=$abl (bits: 2 = no epilogue traffic, 4 = no weight loads, 8 = no patch reads, 16 = no output updates)"
    FACEHIP_W2_ABLATE=$abl FACEHIP_LIB=/tmp/libfacehip_w2prof.so python scripts/wino2_prof.py $shape 2>&1 | grep -v amdgpu.ids
  done
done
