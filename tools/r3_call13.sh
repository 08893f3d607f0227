#!/bin/bash
mkdir -p gpurun_out/r3c13
O=gpurun_out/r3c13
VS_MICRO_WGRAD=0 timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err; grep -E "dy|dil1 act|dil4 act" $O/nhwc_micro.json
timeout 300 python -m pytest tests/test_gpu_nhwc.py -x -q -k "dy" > $O/pytest_dy.log 2>&1; tail -2 $O/pytest_dy.log
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c13/bench_bf16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
PY
