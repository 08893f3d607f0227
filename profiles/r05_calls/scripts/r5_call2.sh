#!/bin/bash
set -u
O=gpurun_out/r5c2
mkdir -p $O
export TMPDIR=/tmp
timeout 100 tools/occupancy_probe > $O/occupancy_probe.txt 2>&1; cat $O/occupancy_probe.txt
timeout 1500 python -m pytest tests/test_gpu_nhwc_f16x3.py tests/test_gpu_b64.py "tests/test_gpu_trainer.py::test_train_step_after_a_plain_autograd_loop_uses_the_fresh_gradients" tests/test_gpu_boundary.py -q --timeout=900 2>&1 | tail -60 > $O/pytest.log
tail -40 $O/pytest.log
