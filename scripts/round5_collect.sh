#!/bin/bash
# Round-5 evidence run (GPU box): rocprofv3 stats + traffic counters for the headline and the detect workload, per-kernel SQ / LDS
# counters, per-layer tables (B = 128 and the C4 batch 64), secondary configurations, the two-stream co-residency
# trace, the headline line.  Everything lands under gpurun_out/ (scratch); scripts/summarize_profile.py + scripts/secondary_table.py
# + scripts/overlap_trace.py turn it into profiles/r05_*.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench done"
bash scripts/profile_round.sh r05 > gpurun_out/r05_profile.log 2>&1; echo "profile e2e done"
bash scripts/profile_round.sh r05det --workload detect > gpurun_out/r05det_profile.log 2>&1; echo "profile detect done"
PYTHONPATH=. python scripts/layer_times.py det 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_layer_times_det128.txt
PYTHONPATH=. python scripts/layer_times.py rec 128 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_layer_times_rec128.txt
PYTHONPATH=. python scripts/layer_times.py rec 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_layer_times_rec64.txt; echo "layer tables done"
bash scripts/secondary_configs.sh > gpurun_out/r05_secondary.log 2>&1; echo "secondary done"
rm -rf gpurun_out/pmck_r05rec_* gpurun_out/pmck_r05det_*
bash scripts/pmc_kernels.sh r05rec --serial > gpurun_out/r05_pmc_rec.log 2>&1; echo "pmc e2e done"
bash scripts/pmc_kernels.sh r05det --workload detect > gpurun_out/r05_pmc_det.log 2>&1; echo "pmc det done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05_overlap -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --sustain-steps 0 --no-kernel-timing > $R/gpurun_out/r05_overlap_bench.json 2> $R/gpurun_out/r05_overlap.err; echo "overlap trace done"
