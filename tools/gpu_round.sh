set -u
export TMPDIR=/tmp
for P in 2 3 4; do
  echo "=== P=$P"
  VS_CONV71_P=$P timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv64" 2>&1 | grep -E "passed|failed|error" | cut -c1-200
  VS_CONV71_P=$P timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stage_ms']; print(d['ms_per_step'], 'cnn2', s['cnn2'], 'dgrad_cnn2', s['dgrad_cnn2'], 'cnn3', s['cnn3'], 'fwd', d['forward'])"
done
