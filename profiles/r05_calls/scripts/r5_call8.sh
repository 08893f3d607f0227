#!/bin/bash
set -u
O=gpurun_out/r5c8
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_boundary.py tests/test_gpu_trainer.py tests/test_gpu_b64.py -q -x --timeout=900 2>&1 | tail -8
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench.err | tail -1 > $O/bench$i.json
python - $O/bench$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print("ms/step",d["ms_per_step"],"cal",d["box_calibration"]["ms"],"fwd_bn",s.get("fwd_bn"),"bwd_bn",s.get("bwd_bn"),"lstm_gemm",s["lstm_gemm"],"bwd_lstm_gemm",s["bwd_lstm_gemm"], "gemm frac", d["roofline"]["lstm_input_gemm"]["frac"])
PY
done
