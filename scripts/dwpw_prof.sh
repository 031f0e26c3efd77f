#!/bin/bash
# diagnostic build of dwpw_mfma.hip with phase stamps into a side copy of the library, then scripts/dwpw_prof.py on the SCRFD shapes
set -e
cd "$(dirname "$0")/.."
B=build/facehip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DFACEHIP_DWPW_PROF -c facerecognizeonnx_amd/csrc/dwpw_mfma.hip -o $B/dwpw_mfma_prof.o
cp facerecognizeonnx_amd/libfacehip.so /tmp/libfacehip_backup.so
# whatever happens below, the production library comes back (a failed run must never leave the diagnostic build installed)
trap 'cp /tmp/libfacehip_backup.so facerecognizeonnx_amd/libfacehip.so' EXIT
OBJS=$(ls $B/*.o | grep -v dwpw_mfma.o | grep -v dwpw_mfma_prof.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o facerecognizeonnx_amd/libfacehip.so $OBJS $B/dwpw_mfma_prof.o -lz
SHAPES=("80 80 72 72" "160 160 40 40" "80 80 64 64" "320 320 16 16" "320 320 16 40 128 2")
if [ -n "$DWPW_PROF_SHAPE" ]; then SHAPES=("$DWPW_PROF_SHAPE"); fi
for shape in "${SHAPES[@]}"; do echo "== $shape"; python scripts/dwpw_prof.py $shape 2>&1 | grep -v amdgpu.ids; done
