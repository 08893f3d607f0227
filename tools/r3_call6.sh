#!/bin/bash
mkdir -p gpurun_out/r3c6
O=gpurun_out/r3c6
timeout 900 python -m pytest tests/test_gpu_nhwc.py -q -k "gemm_bf16" > $O/pytest_gemm.log 2>&1; echo "rc=$?" >> $O/pytest_gemm.log
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
timeout 400 python bench.py --mode forward --conv-math bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fwd_bf16.json 2> $O/bench_fwd_bf16.err
tail -4 $O/pytest_gemm.log; tail -6 $O/pytest_gpu.log; tail -2 $O/bench_train_bf16.err
