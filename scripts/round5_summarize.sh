#!/bin/bash
# Turns the scratch output of scripts/round5_collect.sh (gpurun_out/) into the tracked profiles/r05_* files.
set -e
cd "$(dirname "$0")/.."
python scripts/summarize_profile.py r05 e2e > /dev/null
python scripts/summarize_profile.py r05det detect > /dev/null
python scripts/secondary_table.py r05 > /dev/null
python scripts/pmc_table.py r05rec profiles/r05_pmc_rec128.md "IResNet-50 + SCRFD, headline step (serial form), B = 128" > /dev/null
python scripts/pmc_table.py r05det profiles/r05_pmc_det128.md "SCRFD det_500m, B = 128" > /dev/null
python scripts/overlap_trace.py r05 > /dev/null
cp gpurun_out/r05_bench_line.json profiles/r05_bench_line.json
for f in layer_times_det128 layer_times_rec128 layer_times_rec64; do cp gpurun_out/r05_$f.txt profiles/r05_$f.txt; done
ls profiles/r05*
