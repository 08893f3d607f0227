#!/bin/bash
mkdir -p gpurun_out/r3c4
O=gpurun_out/r3c4
timeout 900 python -m pytest tests/test_gpu_nhwc.py -q > $O/pytest_nhwc.log 2>&1; echo "rc=$?" >> $O/pytest_nhwc.log
timeout 900 python -m pytest tests/test_gpu_bf16.py -q > $O/pytest_bf16.log 2>&1; echo "rc=$?" >> $O/pytest_bf16.log
timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_train_bf16.json 2> $O/bench_train_bf16.err
timeout 300 python bench.py --mode longform --conv-math bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_longform_bf16.json 2> $O/bench_longform_bf16.err
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?" >> $O/bench_gpus2.err
tail -4 $O/pytest_nhwc.log; tail -4 $O/pytest_bf16.log; tail -3 $O/bench_train_bf16.err; tail -3 $O/bench_longform_bf16.err; tail -3 $O/bench_gpus2.err
