cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CYCLES SQ_CYCLES SQ_WAVES SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l$i -- $R/tools/lauum_probe 16384 > /dev/null 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_p$i -- $R/tools/mfma_peak > /dev/null 2>&1
done
find $R/gpurun_out/pmc_* -name "*counter_collection.csv" | head -20
