#!/bin/bash
# where the LSTM leaf contractions start on the side stream (VS_OPT_LSTM_LEAF_LATE 0 / 1 / 2): parity, then A/B of the step
mkdir -p gpurun_out/r5c24
python -m pytest tests/test_gpu_bf16.py -x -q -k "leaf or head_data" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c24/pytest.log
cat gpurun_out/r5c24/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('bwd_head','bwd_lstm_rec','bwd_lstm_gemm','bwd_edge','wgrad_cnn7','dgrad_cnn7')})" | tee -a gpurun_out/r5c24/ab.txt; }
for rep in 1 2 3; do
for mode in 0 1 2; do
TAG="train leaf_late=$mode" VOICESPLIT_LSTM_LEAF_LATE=$mode run
done; done
