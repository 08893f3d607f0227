#!/bin/bash
# after the absmax fix and the test changes: forward tests, forward legs
mkdir -p gpurun_out/r5c26
python -m pytest tests/test_gpu_forward.py tests/test_gpu_nhwc_f16x3.py tests/test_gpu_head.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c26/pytest.log
cat gpurun_out/r5c26/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], {k:v for k,v in d['stage_ms'].items() if v})" | tee -a gpurun_out/r5c26/ab.txt; }
for rep in 1 2; do
TAG="fwd f16x3" run --mode forward
done
