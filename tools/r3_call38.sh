#!/bin/bash
# full GPU suite + smoke + headline bench after the LSTM / trainer / GEMM changes
O=gpurun_out/r3c38
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c38/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if v})
print("roofline", d["roofline"]["frac"], d["roofline"]["lstm_input_gemm"])
print("fwd bf16", d["forward_bf16"]["value"], d["forward_bf16"]["ms_per_step"], "fwd f16x3", d["forward"]["value"], d["forward"]["ms_per_step"], "fp32_class", d["fp32_class"]["value"], "strict", d["fp32_strict"]["value"])
PY
