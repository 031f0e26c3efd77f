#!/bin/bash
# SQ / LDS / memory counters per kernel for one bench workload (runs on the GPU box; counters in their own passes, kernel-trace only).
# Usage: bash scripts/pmc_kernels.sh <tag> <bench.py args...>      -> gpurun_out/pmc_<tag>.md
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --sustain-steps 0 --no-kernel-timing > /dev/null 2> $R/gpurun_out/pmck_${TAG}_$i.err
  echo "pass $i done"
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    n = n.replace("void ", "").replace("fh::", "").replace("(anonymous namespace)::", ""); n = re.sub(r"\(.*", "", n)
    return n[:70]
for f in glob.glob("$R/gpurun_out/pmck_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
        "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "GRBM_GUI_ACTIVE",
        "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]
with open("$R/gpurun_out/pmc_${TAG}.md", "w") as o:
    o.write("# per-kernel counter means (per launch), workload: $*\n\n| kernel | launches | " + " | ".join(cols) + " |\n|---|---|" + "---|" * len(cols) + "\n")
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
        n = max(len(v) for v in acc[k].values())
        o.write(f"| {k} | {n} | " + " | ".join(f"{sum(acc[k][c]) / len(acc[k][c]):.4g}" if acc[k].get(c) else "-" for c in cols) + " |\n")
print(open("$R/gpurun_out/pmc_${TAG}.md").read()[:6000])
PY
