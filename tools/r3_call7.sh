#!/bin/bash
mkdir -p gpurun_out/r3c7
O=gpurun_out/r3c7
VS_MICRO_ONLY=none timeout 300 python tools/gemm_micro.py > $O/gemm_micro.json 2> $O/gemm_micro.err
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o trace -f csv -- python $R/tools/gemm_micro.py > $R/$O/trace.log 2>&1
cd $R
find $O/trace -type f ! -name "*stats.csv" -delete
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_trainer.py -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log; cat $O/gemm_micro.json | tail -30
