#!/bin/bash
mkdir -p gpurun_out/r3c19
O=gpurun_out/r3c19
timeout 300 python tools/wgrad_ablation.py > $O/wgrad_abl.json 2> $O/err.log; cat $O/wgrad_abl.json
timeout 300 python -m pytest tests/test_gpu_nhwc.py -x -q -k "weight_gradient or bn_act or first_bn" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c19/bench_bf16.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
PY
