set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "fused_with_first or module_backward or golden or training_step or layerwise or full_size_module or conv_edge" 2>&1 | tail -5 | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['stage_ms'])"
