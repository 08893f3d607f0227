set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "dgrad_and_wgrad or one_hot" 2>&1 | tail -2 | cut -c1-300
for rep in 1 2; do for m in 2 1; do
VS_MICRO_WGRAD_KERNEL=$m VS_MICRO_REPS=5 timeout 300 python tools/wgrad_micro.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('kernel $m', {k:v['ms'] for k,v in d.items() if isinstance(v,dict) and 'f16' in k})"
done; done
