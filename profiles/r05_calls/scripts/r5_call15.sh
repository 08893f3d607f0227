#!/bin/bash
set -u
O=gpurun_out/r5c15
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lstm16.py tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_trainer.py -q -x --timeout=600 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['ms_per_step'], d['value'], 'bwd_lstm_rec', s['bwd_lstm_rec'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'])"; done
