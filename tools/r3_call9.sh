#!/bin/bash
mkdir -p gpurun_out/r3c9
O=gpurun_out/r3c9
timeout 900 python -m pytest tests/test_gpu_nhwc.py -q > $O/pytest_nhwc.log 2>&1; echo "rc=$?" >> $O/pytest_nhwc.log
VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/gemm_micro.json 2> $O/gemm_micro.err
timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
timeout 600 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
tail -3 $O/pytest_nhwc.log; cat $O/gemm_micro.json; tail -3 $O/pytest_bf16.log
