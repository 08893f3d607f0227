#!/bin/bash
# f16 / bf16 recurrent products + fast gate functions: unit tests, the LSTM-touching module tests, timings, bench A/B
O=gpurun_out/r3c32
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lstm16.py -q > $O/pytest_lstm16.log 2>&1; tail -15 $O/pytest_lstm16.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_forward.py -q -x -k "lstm or bf16 or module or oracle or golden or forward or smoke" > $O/pytest_rest.log 2>&1; tail -8 $O/pytest_rest.log | cut -c1-300
timeout 300 python tools/lstm_time.py 64 > $O/lstm_time.txt 2>&1; timeout 200 python tools/lstm_time.py 2 >> $O/lstm_time.txt 2>&1; grep "B=" $O/lstm_time.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c32/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if "lstm" in k or "head" in k or "edge" in k})
PY
