cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/scripts/conv_bench.py 256 14 14 256 256 3 1 0 20
python $R/scripts/conv_bench.py 334 14 14 256 256 3 1 0 20
python $R/scripts/conv_bench.py 256 56 56 64 64 3 1 1 20
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -- python $R/scripts/conv_bench.py 334 14 14 256 256 3 1 0 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -- python $R/scripts/conv_bench.py 334 14 14 256 256 3 1 0 5 > /dev/null 2>&1
find $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 -name "*.csv" | head
