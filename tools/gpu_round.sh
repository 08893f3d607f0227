set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== lstm tests"; timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -q -k "lstm" 2>&1 | tail -12 | cut -c1-300
