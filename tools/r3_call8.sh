#!/bin/bash
mkdir -p gpurun_out/r3c8
O=gpurun_out/r3c8
timeout 600 python -m pytest tests/test_gpu_nhwc.py -q -k "gemm_bf16" > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/gemm_micro.json 2> $O/gemm_micro.err
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
tail -3 $O/pytest_gemm.log; cat $O/gemm_micro.json; tail -3 $O/pytest_bf16.log
