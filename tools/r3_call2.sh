#!/bin/bash
mkdir -p gpurun_out/r3c2
O=gpurun_out/r3c2
timeout 900 python -m pytest tests/test_gpu_nhwc.py -q > $O/pytest_nhwc.log 2>&1; echo "rc=$?" >> $O/pytest_nhwc.log
timeout 300 python bench.py --mode forward --conv-math bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fwd_bf16.json 2> $O/bench_fwd_bf16.err
timeout 300 python bench.py --mode forward --conv-math bf16 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fwd_bf16_b256.json 2> $O/bench_fwd_bf16_b256.err
timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err
tail -4 $O/pytest_nhwc.log
