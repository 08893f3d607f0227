set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_trainer.py -m gpu -q --tb=short -k "exact_long or rccl" > gpurun_out/pytest_fail.log 2>&1; tail -5 gpurun_out/pytest_fail.log | cut -c1-300
echo "=== bench"; timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err; python -c "
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1]); print(d['metric']); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(d.get('forward')); print(d.get('rccl'))"
