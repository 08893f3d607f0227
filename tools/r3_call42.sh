#!/bin/bash
O=gpurun_out/r3c42
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trainer.py -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for rep in 1 2; do for v in 0 1; do
  VOICESPLIT_LEAF_DEFER=$v timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('defer=$v', d['value'], d['ms_per_step'], {k: v for k, v in d['stage_ms'].items() if k in ('bwd_lstm_gemm','bwd_edge','bwd_bn','dgrad_cnn7','dgrad_cnn6','wgrad_cnn7')})
"
done; done
