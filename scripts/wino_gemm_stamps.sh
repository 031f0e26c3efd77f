#!/bin/bash
# Phase stamps of wino_gemm_kernel (winograd.hip): a diagnostic build (-DFACEHIP_WINO_STAMP) installed as a SIDE copy of the library
# (FACEHIP_LIB), then one Winograd layer through the single-layer test entry point; per workgroup the 100 MHz wall clock at entry, first
# chunk landed, K loop done, last store issued, stores acknowledged.   scripts/wino_gemm_stamps.sh [B H W Cin Cout] ...
set -e
cd "$(dirname "$0")/.."
B=build/facehip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DFACEHIP_WINO_STAMP -c facerecognizeonnx_amd/csrc/winograd.hip -o $B/winograd_stamp.o
OBJS=$(ls $B/*.o | grep -v "/winograd.o" | grep -v "_prof.o" | grep -v "_abl.o" | grep -v "_stamp.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libfacehip_wstamp.so $OBJS $B/winograd_stamp.o -lz -ldl
for shape in "${@:-128 14 14 256 256}"; do
  FACEHIP_LIB=/tmp/libfacehip_wstamp.so PYTHONPATH=. python scripts/wino_gemm_stamps.py $shape 2>&1 | grep -v amdgpu.ids
done
