# Round-end GPU script (one gpurun call): full -m gpu suite, smoke(), the bench lines, both rocprofv3 profiles.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/final_round.sh r02'
set -u
R=${1:-r02}
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2> $O/bench_train.err | tail -1 > $O/${R}_bench_train.json; cut -c1-260 $O/${R}_bench_train.json
timeout 300 python bench.py --mode forward 2> $O/bench_forward.err | tail -1 > $O/${R}_bench_forward.json; cut -c1-200 $O/${R}_bench_forward.json
timeout 300 python bench.py --conv-math bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_train_bf16.json; cut -c1-200 $O/${R}_bench_train_bf16.json
timeout 300 python bench.py --conv-math fp32 --no-cpu-baseline --steps 5 2>/dev/null | tail -1 > $O/${R}_bench_train_fp32math.json; cut -c1-200 $O/${R}_bench_train_fp32math.json
timeout 300 python bench.py --serial-backward --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_train_serial_backward.json; cut -c1-200 $O/${R}_bench_train_serial_backward.json
timeout 300 python bench.py --model voicefilter --loss powerlaw --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_train_voicefilter_powerlaw.json; cut -c1-200 $O/${R}_bench_train_voicefilter_powerlaw.json
timeout 200 python bench.py --batch 2 --steps 30 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_train_b2.json; cut -c1-200 $O/${R}_bench_train_b2.json
bash tools/profile_gpu.sh ${R}_train 2>&1 | tail -2
bash tools/profile_gpu.sh ${R}_forward --mode forward 2>&1 | tail -2
