set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_backward.py -m gpu -q -k "golden or module or full" 2>&1 | tail -3 | cut -c1-400
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_q.json'))
print(d['value'], d['ms_per_step']); print({k:v for k,v in d['stage_ms'].items() if v})
PY
