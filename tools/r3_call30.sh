#!/bin/bash
# validate HEAD (bf16 head GEMM instances) : full GPU suite + headline bench
mkdir -p gpurun_out/r3c30
O=gpurun_out/r3c30
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c30/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
print(d["roofline"]["frac"], d["forward_bf16"]["value"], d["forward"]["value"])
PY
