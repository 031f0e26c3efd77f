#!/bin/bash
# Runs on the GPU box: kernel-trace stats + HBM traffic counters for the default bench workload.
# Usage: bash scripts/profile_round.sh r01 [bench.py args, e.g. --workload detect]
TAG=${1:-r01}
shift || true
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# stamp: which sources these counters were measured on (bench.py drops roofline.traffic when csrc/ has changed since)
(cd $R && python3 -c "import json,time,sys; sys.path.insert(0,'.'); from facerecognizeonnx_amd._lib import csrc_fingerprint as f; print(json.dumps({'csrc_sha16': f(), 'date': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())}))" > $OUT/stamp.json)
cd /tmp && export TMPDIR=/tmp
# --serial on the stats leg: with the default two-stream streaming form the detector and recogniser kernels time-share the chip, which
# stretches every kernel's duration; the roofline's avg launch duration (bench.py's instrumented pass) is of the kernel alone.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain-steps 0 --serial "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-steps 0 --no-kernel-timing "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-steps 0 --no-kernel-timing "$@" > /dev/null 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
