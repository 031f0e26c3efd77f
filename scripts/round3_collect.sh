#!/bin/bash
# Round-3 evidence run (GPU box): rocprofv3 stats + traffic counters for the headline and the detect workload, per-layer tables,
# secondary configurations, the latency line, the dwpw phase stamps.  Everything lands under gpurun_out/ (scratch); the summaries are
# copied into profiles/ by scripts/summarize_profile.py + by hand.
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/profile_round.sh r03 > gpurun_out/r03_profile.log 2>&1
echo "profile e2e done"
bash scripts/profile_round.sh r03det --workload detect > gpurun_out/r03det_profile.log 2>&1
echo "profile detect done"
PYTHONPATH=. python scripts/layer_times.py det 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_layer_times_det128.txt
PYTHONPATH=. python scripts/layer_times.py rec 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_layer_times_rec128.txt
echo "layer tables done"
bash scripts/secondary_configs.sh > gpurun_out/r03_secondary.log 2>&1
echo "secondary done"
bash scripts/dwpw_prof.sh > gpurun_out/r03_dwpw_phases.txt 2>&1
echo "phases done"
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
echo "bench done"
