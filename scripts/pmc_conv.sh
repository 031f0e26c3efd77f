# MFMA-pipe / wait counters of the dominant conv configuration on a shape with whole tile rounds (B=334: 1024 tiles).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/scripts/conv_bench.py 334 14 14 256 256 3 1 0 20
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_WAVES"; do
  d=$R/gpurun_out/pmc_$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/scripts/conv_bench.py 334 14 14 256 256 3 1 0 5 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
