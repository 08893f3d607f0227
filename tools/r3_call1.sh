#!/bin/bash
# round-3 GPU call 1: hardware probes, channels-last conv parity + timing, GEMM rasterization A/B
mkdir -p gpurun_out/r3c1
O=gpurun_out/r3c1
timeout 60 ./tools/probe_cdna4 > $O/probe.log 2>&1; echo "probe rc=$?" >> $O/probe.log
timeout 600 python -m pytest tests/test_gpu_nhwc.py -x -q > $O/pytest_nhwc.log 2>&1; echo "rc=$?" >> $O/pytest_nhwc.log
timeout 300 python tools/nhwc_micro.py > $O/nhwc_micro.json 2> $O/nhwc_micro.err
for band in 0 8 4; do
  VOICESPLIT_GEMM_BAND=$band timeout 300 python bench.py --mode forward --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fwd_band$band.json 2> $O/bench_fwd_band$band.err
done
VOICESPLIT_GEMM_BAND=0 VS_MICRO_ONLY= timeout 200 python tools/gemm_micro.py > $O/gemm_micro_band0.json 2>&1
VOICESPLIT_GEMM_BAND=8 timeout 200 python tools/gemm_micro.py > $O/gemm_micro_band8.json 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_forward.py -x -q > $O/pytest_fwd.log 2>&1; echo "rc=$?" >> $O/pytest_fwd.log
tail -5 $O/probe.log $O/pytest_nhwc.log $O/pytest_fwd.log
