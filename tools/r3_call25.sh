#!/bin/bash
mkdir -p gpurun_out/r3c25
O=gpurun_out/r3c25
timeout 600 python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_bf16.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c25/bench_bf16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
PY
