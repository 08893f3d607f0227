#!/bin/bash
set -u
O=gpurun_out/r5c6
mkdir -p $O
export TMPDIR=/tmp
VS_MICRO_LIB=libvoicesplit_hip_probe.so VS_MICRO_CONV8_PROBE=1 timeout 300 python tools/nhwc_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/conv8_probe.txt
