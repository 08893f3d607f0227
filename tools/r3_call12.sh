#!/bin/bash
# dy-fusion: unit tests, whole-path bf16 tests, training step time and per-stage profile
mkdir -p gpurun_out/r3c12
O=gpurun_out/r3c12
timeout 600 python -m pytest tests/test_gpu_nhwc.py -x -q -k "dy or cnn8 or cnn1 or bn_act" > $O/pytest_dy.log 2>&1; tail -5 $O/pytest_dy.log
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_nhwc.py -x -q -k "not dy_epilogue and not matches_fp64 and not gemm" > $O/pytest_bf16.log 2>&1; tail -5 $O/pytest_bf16.log
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -1 $O/bench_bf16.json | cut -c1-600
timeout 600 python bench.py --conv-math bf16 --no-extras --steps 5 --warmup 2 --serial-backward > $O/bench_bf16_serial.json 2> $O/bench_bf16_serial.err; tail -3 $O/bench_bf16_serial.err
python - <<'PY'
import json
for f in ("bench_bf16", "bench_bf16_serial"):
    d = json.loads(open(f"gpurun_out/r3c12/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
PY
