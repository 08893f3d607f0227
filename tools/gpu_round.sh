set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest"; timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-400
echo "=== bench"; timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err; python -c "
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:round(v,2) for k,v in d['stage_ms'].items() if v}); print(d.get('forward'))"
