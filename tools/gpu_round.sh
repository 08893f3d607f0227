set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "dgrad_and_wgrad or one_hot" 2>&1 | tail -6 | cut -c1-600
for m in 2 1 0; do
VS_MICRO_WGRAD_KERNEL=$m timeout 300 python tools/wgrad_micro.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('kernel $m', {k:v['ms'] for k,v in d.items() if isinstance(v,dict) and 'f16' in k})"
done
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:v for k,v in d['stage_ms'].items() if 'wgrad' in k})"
