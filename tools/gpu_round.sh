set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/train_step_timing.py --steps 2 --warmup 1 > gpurun_out/train_timing.log 2>&1; cat gpurun_out/train_timing.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err; tail -3 gpurun_out/bench_train.err; cut -c1-400 gpurun_out/bench_train.json
