#!/bin/bash
# the loss value read from pinned memory (copied beside the backward) instead of the blocking loss.item() behind optimizer.step(): test + A/B
mkdir -p gpurun_out/r5c28
python -m pytest tests/test_gpu_trainer.py tests/test_gpu_bf16.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head -20 > gpurun_out/r5c28/pytest.log
cat gpurun_out/r5c28/pytest.log
run() { python bench.py "$@" --no-extras --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c28/ab.txt; }
run_long() { python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r5c28/ab.txt; }
for rep in 1 2 3; do
TAG="train early_loss=0" VOICESPLIT_EARLY_LOSS=0 run
TAG="train early_loss=1" run
done
TAG="train early_loss=0 steps=40" VOICESPLIT_EARLY_LOSS=0 run_long
TAG="train early_loss=1 steps=40" run_long
