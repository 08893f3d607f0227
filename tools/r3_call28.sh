#!/bin/bash
mkdir -p gpurun_out/r3c28
O=gpurun_out/r3c28
timeout 300 python -m pytest tests/test_gpu_nhwc.py -x -q -k "cnn1 or first or cnn8" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
VS_MICRO_WGRAD=0 VS_MICRO_DY=0 timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err; python - <<'PY'
import json
for k, v in json.load(open("gpurun_out/r3c28/nhwc_micro.json")).items():
    if "dil4" in k or "7x1" in k: print(k, v)
PY
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c28/bench_bf16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
PY
