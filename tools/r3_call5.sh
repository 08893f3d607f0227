#!/bin/bash
mkdir -p gpurun_out/r3c5
O=gpurun_out/r3c5
timeout 400 python bench.py --conv-math bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --serial-backward > $O/bench_train_bf16_serial.json 2> $O/bench_train_bf16_serial.err
bash tools/profile_gpu.sh r03_train_bf16 --conv-math bf16 > $O/profile.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
