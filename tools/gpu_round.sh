set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py -m gpu -q -x -k "bn_act or layerwise or train or golden or oracle" 2>&1 | tail -5 | cut -c1-300
timeout 300 python tools/train_step_timing.py 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1
