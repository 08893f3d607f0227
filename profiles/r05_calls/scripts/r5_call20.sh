#!/bin/bash
# fused cnn8 -> GEMM operand: parity, then A/B of the XCD-aware block order in the cnn8 kernels (VOICESPLIT_GEMM_ABL=99: off)
mkdir -p gpurun_out/r5c20
python -m pytest tests/test_gpu_forward.py tests/test_gpu_lstm16.py tests/test_gpu_nhwc.py tests/test_gpu_nhwc_f16x3.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c20/pytest.log
cat gpurun_out/r5c20/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('cnn1','cnn8','lstm_gemm')})" | tee -a gpurun_out/r5c20/ab.txt; }
for rep in 1 2; do
TAG="fwd f16x3 rows=0 map=on " VOICESPLIT_FEAT_ROWS=0 run --mode forward
TAG="fwd f16x3 rows=0 map=off" VOICESPLIT_FEAT_ROWS=0 VOICESPLIT_GEMM_ABL=99 run --mode forward
TAG="fwd f16x3 rows=1 map=on " run --mode forward
TAG="fwd f16x3 rows=1 map=off" VOICESPLIT_GEMM_ABL=99 run --mode forward
TAG="fwd bf16 map=on " run --mode forward --conv-math bf16
TAG="fwd bf16 map=off" VOICESPLIT_GEMM_ABL=99 run --mode forward --conv-math bf16
TAG="train map=on " run
TAG="train map=off" VOICESPLIT_GEMM_ABL=99 run
done
