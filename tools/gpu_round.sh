set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
VOICESPLIT_CONV_MATH=f16x3 timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py -m gpu -q -x -k "golden or module or stages or properties" 2>&1 | tail -4 | cut -c1-400
timeout 600 python bench.py --steps 5 --warmup 1 --conv-math f16x3 --no-cpu-baseline > gpurun_out/bench_train_f16x3.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_train_f16x3.json'))
print(d['value'], d['ms_per_step']); print({k:v for k,v in d['stage_ms'].items() if v})
PY
timeout 600 python bench.py --mode forward --steps 5 --warmup 1 --conv-math f16x3 --no-cpu-baseline > gpurun_out/bench_fwd_f16x3.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_fwd_f16x3.json'))
print(d['value'], d['ms_per_step']); print({k:v for k,v in d['stage_ms'].items() if v})
PY
