#!/bin/bash
set -u
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py 2> $O/bench_train.err | tail -1 > $O/r05_bench_train.json; cut -c1-200 $O/r05_bench_train.json
timeout 300 python bench.py --mode forward --no-cpu-baseline 2> $O/bench_forward.err | tail -1 > $O/r05_bench_forward.json; cut -c1-200 $O/r05_bench_forward.json
