mkdir -p gpurun_out/r04
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VRP_GEMM_VARIANT=rows
for v in 1 0; do
  if [ $v = 1 ]; then export VRP_GEMM_ROWS_V1=1; else unset VRP_GEMM_ROWS_V1; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04/pmc_gemm_v$v -o p -- python3 tools/gemm_one.py 81920 384 128 > gpurun_out/r04/pmc_gemm_v$v.log 2>&1
  python3 tools/pmc_sum.py gpurun_out/r04/pmc_gemm_v$v gemm_rows
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/r04/pmc_gemm2_v$v -o p -- python3 tools/gemm_one.py 81920 384 128 > gpurun_out/r04/pmc_gemm2_v$v.log 2>&1
  python3 tools/pmc_sum.py gpurun_out/r04/pmc_gemm2_v$v gemm_rows
done
