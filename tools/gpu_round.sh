set -u
export TMPDIR=/tmp
echo "=== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | cut -c1-300
echo "=== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/r02_bench_train.json | cut -c1-400
