set -u
export TMPDIR=/tmp
echo "=== lstm tests"; timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "persistent_lstm or bilstm" 2>&1 | grep -E "passed|failed|error|Error|assert" | cut -c1-400
echo "=== timing"; timeout 300 python tools/lstm_time.py 64 2>&1 | grep "B="; timeout 300 python tools/lstm_time.py 2 2>&1 | grep "B="
