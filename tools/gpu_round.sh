set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
VOICESPLIT_CONV_MATH=f16x3 timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/t_all_f16x3.log
grep -E "^E  |passed|failed|skipped" gpurun_out/t_all_f16x3.log | cut -c1-600
timeout 600 python bench.py --steps 5 --warmup 1 --conv-math f16x3 --no-cpu-baseline > gpurun_out/bench_train_f16x3.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench_train_f16x3.json
timeout 300 python bench.py --mode forward --steps 5 --warmup 1 --conv-math f16x3 --no-cpu-baseline > gpurun_out/bench_fwd_f16x3.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench_fwd_f16x3.json
