#!/bin/bash
# gradient sink / fused Adam / host prologue / absmax skip: trainer + bf16 + lstm16 tests, headline bench, step timeline
O=$PWD/gpurun_out/r3c33
mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_lstm16.py tests/test_gpu_bf16.py -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_train.json 2> $O/bench_train.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c33/bench_train.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print({k: v for k, v in d["stage_ms"].items() if "lstm" in k or "head" in k or "edge" in k})
PY
timeout 300 python bench.py --batch 2 --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace -f csv -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $REPO/tools/step_timeline.py $F > $O/timeline_b64.txt; head -12 $O/timeline_b64.txt
find $O/trace -type f -delete
