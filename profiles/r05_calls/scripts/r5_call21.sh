#!/bin/bash
# head data gradients on gemm_bf16 (VS_OPT_HEAD_BWD_GEMM): parity, then A/B of the training step
mkdir -p gpurun_out/r5c21
python -m pytest tests/test_gpu_boundary.py tests/test_gpu_bf16.py tests/test_gpu_forward.py tests/test_gpu_trainer.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c21/pytest.log
cat gpurun_out/r5c21/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('bwd_head','bwd_lstm_rec','bwd_lstm_gemm')})" | tee -a gpurun_out/r5c21/ab.txt; }
for rep in 1 2 3; do
TAG="train head_bwd_gemm=0" VOICESPLIT_HEAD_BWD_GEMM=0 run
TAG="train head_bwd_gemm=1" run
done
