#!/bin/bash
# PMC passes for the GEMM micro-benchmark (run on the GPU box): writes CSVs under gpurun_out/pmc_*
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" ; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/bench_gemm.py > $R/gpurun_out/pmc_$tag.log 2>&1
done
ls -R $R/gpurun_out | grep -i csv | head
