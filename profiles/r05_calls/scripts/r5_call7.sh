#!/bin/bash
set -u
O=gpurun_out/r5c7
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_nhwc.py -q -x --timeout=600 -k "gemm" 2>&1 | tail -5
VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/gemm_micro.json 2>/dev/null; cat $O/gemm_micro.json | head -60
timeout 200 python tools/gemm_epilogue_probe.py > $O/gemm_epilogue_probe.json 2>/dev/null; cat $O/gemm_epilogue_probe.json | head -40
