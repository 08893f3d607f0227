#!/bin/bash
set -u
O=gpurun_out/r5c12
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lstm16.py -q -x --timeout=600 2>&1 | tail -6
timeout 300 python tools/lstm_time.py 64 2>/dev/null | grep "B=" | tee $O/lstm_time.txt
timeout 300 python tools/lstm_time.py 2 2>/dev/null | grep "B=" | tee -a $O/lstm_time.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_forward.py tests/test_gpu_trainer.py -q -x --timeout=600 2>&1 | tail -4
