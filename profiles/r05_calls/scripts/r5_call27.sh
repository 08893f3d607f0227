#!/bin/bash
# LSTM leaves: dW_ih (the persistent contraction) behind the small leaves (VS_OPT_LSTM_LEAF_LATE=3) vs first (0)
mkdir -p gpurun_out/r5c27
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('bwd_lstm_gemm','bwd_edge','bwd_bn','dgrad_cnn7')})" | tee -a gpurun_out/r5c27/ab.txt; }
for rep in 1 2 3; do
for mode in 0 3; do
TAG="train leaf_late=$mode" VOICESPLIT_LSTM_LEAF_LATE=$mode run
done; done
