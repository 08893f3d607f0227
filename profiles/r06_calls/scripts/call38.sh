# the round's closing call: full GPU suite on the final library, the deterministic-mode line again (its cnn8 grid back at 2048), the default line again
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/r06_pytest_gpu.log; cat $O/r06_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
VOICESPLIT_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/r06_bench_train_deterministic.json; cut -c1-200 $O/r06_bench_train_deterministic.json
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/r06_bench_train_again.json; cut -c1-200 $O/r06_bench_train_again.json
