#!/bin/bash
# head backward: bias column sums to the side stream, dlogits' bf16 rows from the sigmoid-gradient kernel; parity, then A/B of the step
mkdir -p gpurun_out/r5c22
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_trainer.py tests/test_gpu_head.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c22/pytest.log
cat gpurun_out/r5c22/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('bwd_head','bwd_lstm_rec','bwd_lstm_gemm')})" | tee -a gpurun_out/r5c22/ab.txt; }
for rep in 1 2 3; do
TAG="train head_bwd_gemm=0" VOICESPLIT_HEAD_BWD_GEMM=0 run
TAG="train head_bwd_gemm=1" run
TAG="train head_bwd_gemm=1 leaf_side=0" VOICESPLIT_HEAD_LEAF_SIDE=0 run
done
