set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_loss.py -m gpu -q 2>&1 | tail -12 | cut -c1-400
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_q.json'))
print(d['value'], d['ms_per_step'], sum(v for v in d['stage_ms'].values() if v))
PY
