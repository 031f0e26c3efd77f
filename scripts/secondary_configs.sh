#!/bin/bash
# Runs the other BASELINE.json configurations on the GPU box; one JSON line each into gpurun_out/secondary/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/secondary; mkdir -p $O
run() { name=$1; shift; python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || echo "FAILED $name"; }
run headline
run c2_embed --workload embed
run c3_detect --workload detect
run f4 --faces-per-frame 4
run c4_gallery --gallery 1000000 --frames 64
run c4_match --workload match --gallery 1000000 --queries 64 --topk 16
run c5_rank --workload match --gallery 10000000 --as-rank 7 --of-world 8 --queries 64 --topk 16
run from_host --from-host
run serial --serial
run latency --workload latency
run mbf_embed --workload embed --recogniser mbf
run mbf_e2e --recogniser mbf
ls $O
