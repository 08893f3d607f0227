#!/bin/bash
O=gpurun_out/r3c39
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_nhwc.py -q -k "trainer or last or cnn8 or edge or bf16_path or stack" > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c39/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if "lstm" in k or "head" in k or "edge" in k or "cnn1" in k or "cnn8" in k})
PY
