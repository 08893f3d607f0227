#!/bin/bash
# fused cnn8 -> GEMM operand: parity, then A/B of the forward leg
mkdir -p gpurun_out/r5c19
python -m pytest tests/test_gpu_forward.py tests/test_gpu_lstm16.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c19/pytest.log
cat gpurun_out/r5c19/pytest.log
for rep in 1 2; do
for mode in 0 1; do
VOICESPLIT_FEAT_ROWS=$mode python bench.py --mode forward --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('feat_rows=$mode', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('cnn7','cnn8','lstm_gemm')})" | tee -a gpurun_out/r5c19/ab.txt
done; done
