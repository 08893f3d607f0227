#!/bin/bash
# stability of the final defaults over a long run: 300 training steps, then 100 forward steps of each arithmetic
mkdir -p gpurun_out/r5c29
run() { python bench.py "$@" --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'], d['steps'])" | tee -a gpurun_out/r5c29/long.txt; }
TAG="train 10 steps" run --steps 10
TAG="train 300 steps" run --steps 300
TAG="train 10 steps" run --steps 10
TAG="forward f16x3 100 steps" run --mode forward --steps 100
TAG="forward bf16 100 steps" run --mode forward --conv-math bf16 --steps 100
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2 | tee -a gpurun_out/r5c29/long.txt
