#!/bin/bash
mkdir -p gpurun_out/r3c11
O=gpurun_out/r3c11
for cfg in 000 111 222 000 222; do
  VOICESPLIT_GEMM_DMA=$cfg VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/gemm_micro_$cfg.json 2>> $O/gemm_micro.err
  echo "== $cfg"; grep -A2 "bf16 gemm" $O/gemm_micro_$cfg.json | grep -v tflops | tr -d '\n'; echo
done
