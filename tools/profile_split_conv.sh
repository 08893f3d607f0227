#!/bin/bash
# rocprofv3 passes over tools/split_conv_micro.py (the channels-last split-f16 conv alone).  Usage: tools/profile_split_conv.sh <tag>
set -u
TAG=${1:-split}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
CMD="python $REPO/tools/split_conv_micro.py prof_$TAG/micro"
K="nhwc_conv_f16x3_kernel"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -f csv -- $CMD > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc -f csv -- $CMD > "$OUT/pmc_sq.log" 2>&1; echo "pmc_sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -f csv -- $CMD > "$OUT/pmc_fetch.log" 2>&1; echo "pmc_fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -f csv -- $CMD > "$OUT/pmc_write.log" 2>&1; echo "pmc_write rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM -d "$OUT/pmc_lds" -o pmc -f csv -- $CMD > "$OUT/pmc_lds.log" 2>&1; echo "pmc_lds rc=$?"
timeout 300 rocprofv3 --kernel-trace --kernel-include-regex "$K" --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_tcc" -o pmc -f csv -- $CMD > "$OUT/pmc_tcc.log" 2>&1; echo "pmc_tcc rc=$?"
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "*.json" -delete
find "$OUT" -name "*agent_info*" -delete
du -sh "$OUT"
