set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "dgrad_and_wgrad or one_hot or layerwise" 2>&1 | tail -3 | cut -c1-300
timeout 300 python tools/wgrad_micro.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:v['ms'] for k,v in d.items() if isinstance(v,dict)})"
