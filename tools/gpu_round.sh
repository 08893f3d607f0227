set -u
export TMPDIR=/tmp
echo "=== forward tests"; timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | cut -c1-400
echo "=== latency"; timeout 300 python tools/eval_latency.py 2>&1 | grep "B="
