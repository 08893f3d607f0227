set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py -m gpu -q -x -k "conv64 or dgrad or one_hot" 2>&1 | tail -2 | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:round(v,2) for k,v in d['stage_ms'].items() if v and ('cnn' in k) and ('wgrad' not in k)})"
