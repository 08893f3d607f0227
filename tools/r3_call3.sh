#!/bin/bash
mkdir -p gpurun_out/r3c3
O=gpurun_out/r3c3
timeout 900 python -m pytest tests/test_gpu_nhwc.py -q -x > $O/pytest_nhwc.log 2>&1; echo "rc=$?" >> $O/pytest_nhwc.log
timeout 900 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
tail -4 $O/pytest_nhwc.log; tail -4 $O/pytest_bf16.log; tail -3 $O/bench_train_bf16.err
