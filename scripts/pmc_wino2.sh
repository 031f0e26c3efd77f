#!/bin/bash
# SQ / LDS counters of wino2_kernel on the IResNet stage-1 shapes (GPU box; counters in their own passes, kernel-trace only).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcw2_$i -- python3 $R/scripts/wino2_bench.py 128 > /dev/null 2> $R/gpurun_out/pmcw2_$i.err
  echo "pass $i done"
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmcw2_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("fh::", "").replace("(anonymous namespace)::", "")[:60]
        if "wino2" in k or "conv_tall" in k:
            acc[k + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$R/gpurun_out/pmc_wino2.md", "w") as o:
    for k in sorted(acc):
        o.write(f"## {k}\n")
        for c in sorted(acc[k]):
            v = acc[k][c]; o.write(f"  {c:32s} {sum(v)/len(v):14.5g}  (n={len(v)})\n")
print(open("$R/gpurun_out/pmc_wino2.md").read()[:7000])
PY
