set -u
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err; tail -c 300 $O/bench_train.json
timeout 300 python bench.py --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; tail -c 200 $O/bench_forward.json
timeout 300 python bench.py --conv-math fp32 --no-cpu-baseline > $O/bench_train_fp32math.json 2>/dev/null
timeout 300 python bench.py --model voicefilter --loss powerlaw --no-cpu-baseline > $O/bench_train_voicefilter_powerlaw.json 2>/dev/null
bash tools/profile_gpu.sh r01_train 2>&1 | tail -2
bash tools/profile_gpu.sh r01_forward --mode forward 2>&1 | tail -2
