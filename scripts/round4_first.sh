#!/bin/bash
# Round-4 first GPU call: the new tests, the baseline layer tables, per-kernel PMC tables of the serial headline / detect workloads.
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/r4_tests_new.log 2>&1; echo "new tests rc=$?"
tail -5 gpurun_out/r4_tests_new.log
PYTHONPATH=. python scripts/layer_times.py rec 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_base_layer_rec128.txt; echo "rec table done"
PYTHONPATH=. python scripts/layer_times.py det 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_base_layer_det128.txt; echo "det table done"
bash scripts/pmc_kernels.sh r04rec --serial > gpurun_out/r4_pmc_rec.log 2>&1; echo "pmc e2e done"
bash scripts/pmc_kernels.sh r04det --workload detect > gpurun_out/r4_pmc_det.log 2>&1; echo "pmc det done"
