set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | cut -c1-300
VOICESPLIT_CONV_MATH=fp32 timeout 1500 python -m pytest tests -m gpu -q -k "golden or module or stages or properties or semantics" 2>&1 | tail -3 | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cut -c1-300 gpurun_out/bench_default.json
timeout 600 python bench.py --mode forward > gpurun_out/bench_forward.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cut -c1-300 gpurun_out/bench_forward.json
timeout 600 python bench.py --conv-math fp32 --no-cpu-baseline > gpurun_out/bench_fp32.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cut -c1-300 gpurun_out/bench_fp32.json
