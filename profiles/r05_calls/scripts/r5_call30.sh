#!/bin/bash
# re-test of the one-launch BatchNorm finalize in the BACKWARD (VS_OPT_BN_FUSED_FINALIZE=2) now that the pass beside the weight gradient runs one block per CU
mkdir -p gpurun_out/r5c30
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v and k in ('bwd_bn','wgrad_cnn5','dgrad_cnn5')})" | tee -a gpurun_out/r5c30/ab.txt; }
for rep in 1 2 3; do
TAG="train bn_fused_finalize=1" run
TAG="train bn_fused_finalize=2" VOICESPLIT_BN_FUSED_FINALIZE=2 run
done
