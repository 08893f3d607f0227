set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/t_all.log
grep -E "^E  |passed|failed|SKIP|skipped" gpurun_out/t_all.log | cut -c1-700
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; tail -3 gpurun_out/bench_train.err; cat gpurun_out/bench_train.json
timeout 300 python bench.py --mode forward --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fwd.json 2> gpurun_out/bench_fwd.err; tail -3 gpurun_out/bench_fwd.err; cat gpurun_out/bench_fwd.json
