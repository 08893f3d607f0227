#!/bin/bash
mkdir -p gpurun_out/r3c10
O=gpurun_out/r3c10
VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/gemm_micro.json 2> $O/gemm_micro.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
cat $O/gemm_micro.json; tail -3 $O/bench_default.err
