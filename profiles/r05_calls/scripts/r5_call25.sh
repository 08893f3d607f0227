#!/bin/bash
# bf16 eval forward: cnn8 writes the bf16 rows itself (VS_OPT_FEAT_ROWS); parity, then A/B of the bf16 forward leg
mkdir -p gpurun_out/r5c25
python -m pytest tests/test_gpu_forward.py tests/test_gpu_bf16.py tests/test_gpu_audio.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c25/pytest.log
cat gpurun_out/r5c25/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('cnn8','lstm_gemm','head')})" | tee -a gpurun_out/r5c25/ab.txt; }
for rep in 1 2 3; do
TAG="fwd bf16 rows=0" VOICESPLIT_FEAT_ROWS=0 run --mode forward --conv-math bf16
TAG="fwd bf16 rows=1" run --mode forward --conv-math bf16
done
TAG="train" run
