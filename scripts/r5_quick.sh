#!/bin/bash
# quick status of the three lines that matter (headline, C4, C3): scripts/r5_quick.sh TAG
tag=${1:-q}
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_$tag.json 2> gpurun_out/r5_bench_$tag.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gallery 1000000 --frames 64 > gpurun_out/r5_c4_$tag.json 2> gpurun_out/r5_c4_$tag.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload detect > gpurun_out/r5_c3_$tag.json 2> gpurun_out/r5_c3_$tag.err
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("r5_bench_", "r5_c4_", "r5_c3_"):
    try:
        d = json.loads(open("gpurun_out/%s%s.json" % (f, tag)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    r = d["roofline"]
    print(f + tag, round(d["value"]), "per s;", round(d["ms_per_step"], 3), "ms; serial", (d.get("serial_reference") or {}).get("ms_per_step"),
          "|", r["kernel"], round(r["frac"], 3), "| all conv", (r.get("all_conv_igemm") or {}).get("frac"))
PY
