#!/bin/bash
mkdir -p gpurun_out/r3c22
O=gpurun_out/r3c22
timeout 300 python -m pytest tests/test_gpu_nhwc.py -x -q -k "weight_gradient" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/wgrad_ablation.py > $O/wgrad_abl.json 2> $O/err.log; cat $O/wgrad_abl.json
