set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
REPO=$PWD
cd /tmp
CMD="python $REPO/tools/gemm_micro.py"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_f16x3" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $OUT/a -o pmc -f csv -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_f16x3" --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/b -o pmc -f csv -- $CMD > $OUT/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "gemm_f16x3" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/c -o pmc -f csv -- $CMD > $OUT/c.log 2>&1
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
find $OUT -name "*agent_info*" -delete
du -sh $OUT
