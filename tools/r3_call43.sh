#!/bin/bash
O=gpurun_out/r3c43
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trainer.py tests/test_gpu_backward.py -q -k "bf16 or trainer or pipeline or side_stream or golden" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest.log | head
for rep in 1 2; do for v in "--serial-forward" ""; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras $v 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', d['value'], d['ms_per_step'], {k: v for k, v in d['stage_ms'].items() if k in ('cnn2','cnn3','cnn4','cnn7','fwd_bn','cnn8')})
"
done; done
