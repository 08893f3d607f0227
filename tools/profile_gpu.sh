#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel trace + PMC passes of
# bench.py (training step by default).  Counters are collected in their own runs (--pmc with
# --kernel-trace only).   Usage: tools/profile_gpu.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
shift || true
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# which kernel sources these counters belong to: bench.py reports a committed `traffic` only while the kernel's source file is unchanged
(cd "$REPO/voicesplit_amd/csrc" && sha256sum *.hip *.h *.inc) > "$OUT/sources.sha256"
# (ten steps: the first launch of a kernel in a process runs up to 2x long -- cold instruction cache, first touch of its operands -- and
# weighed 1/20 in the four-step tables of rounds 2-5; 1/50 now)
BENCH="python $REPO/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-extras $*"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace -f csv -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
ONE="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $*"
KREG="conv64_mfma|conv64_wgrad|conv64_f16|gemm_f16x3|gemm_pre|gemm_bf16|lstm16|lstm_persistent|lstm_bwd_persistent|nhwc_conv_kernel|nhwc_conv_f16x3|nhwc_wgrad"
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$KREG" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o pmc -f csv -- $ONE > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$KREG|conv_first|conv_last|bn_|nhwc_" --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -f csv -- $ONE > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$KREG|conv_first|conv_last|bn_|nhwc_" --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -f csv -- $ONE > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$KREG" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM -d "$OUT/pmc_lds" -o pmc -f csv -- $ONE > "$OUT/pmc_lds.log" 2>&1
echo "pmc_lds rc=$?"
find "$OUT" -name "*.csv" | head -40
# keep the merged-back payload small: drop everything but csv/log
find "$OUT" -type f ! -name "*.csv" ! -name "*.log" ! -name "sources.sha256" -delete
find "$OUT" -name "*agent_info*" -delete
du -sh "$OUT"
