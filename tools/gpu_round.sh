set -u
timeout 600 python -m pytest tests/test_gpu_audio.py -m gpu -q 2>&1 | grep -E "^E|passed|failed" | head -10 | cut -c1-300
