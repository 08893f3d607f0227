#!/bin/bash
set -u
O=gpurun_out/r5c16
mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "passed|failed|rror" | tail -8 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py 2> $O/bench_train.err | tail -1 > $O/bench_train.json; cut -c1-400 $O/bench_train.json
