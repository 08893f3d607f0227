set -u
export TMPDIR=/tmp
echo "=== gemm / lstm / forward tests"; timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -m gpu -q -x -k "gemm or bilstm or module or golden or stages or prepared or bf16" 2>&1 | grep -E "passed|failed|error" | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --steps 6 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stage_ms']; print(d['value'], d['ms_per_step'], 'lstm_gemm', s['lstm_gemm'], 'fwd', d['forward']['value'])"
