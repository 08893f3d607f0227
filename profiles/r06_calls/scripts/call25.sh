# fp32-class backward: the two LSTM-input contractions on pre-split row-form operands (gemm_pre_kernel) -- parity tests, step, kernel table
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c25; mkdir -p $O
timeout 1500 python -m pytest -q -x --timeout=900 tests/test_gpu_backward.py tests/test_gpu_trainer.py 2>&1 | tail -8
for r in 1 2; do
timeout 300 python bench.py --conv-math f16x3 --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f16x3 step', d['ms_per_step'], d['value'], {k: d['stage_ms'][k] for k in ('bwd_lstm_gemm','lstm_gemm','bwd_bn')})"
done 2>&1 | tee $O/step.txt
bash tools/profile_gpu.sh r06_train_f16x3 --conv-math f16x3 2>&1 | tail -2
