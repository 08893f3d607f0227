#!/bin/bash
O=gpurun_out/r3c41
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_nhwc.py tests/test_gpu_trainer.py tests/test_gpu_backward.py -q -k "bf16 or nhwc or cnn8 or trainer or golden" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c41/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if "lstm" in k or "head" in k or "edge" in k or "cnn1" in k or "cnn8" in k or "fwd_bn" in k})
PY
