set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_audio.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-600
