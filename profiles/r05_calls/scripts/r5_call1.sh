#!/bin/bash
# round 5, GPU call 1: the new parity tests, the ADVICE regression test, and the packed-vs-scalar epilogue A/B
set -u
O=gpurun_out/r5c1
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_nhwc_f16x3.py tests/test_gpu_b64.py "tests/test_gpu_trainer.py::test_train_step_after_a_plain_autograd_loop_uses_the_fresh_gradients" "tests/test_gpu_nhwc.py::test_scalar_and_packed_epilogue_builds_agree_bit_for_bit" tests/test_gpu_boundary.py -q -x --timeout=900 2>&1 | tail -40 > $O/pytest.log
tail -25 $O/pytest.log
VS_MICRO_EPILOGUE_AB=1 VS_MICRO_WGRAD=0 timeout 600 python tools/nhwc_micro.py > $O/nhwc_micro_ab.json 2> $O/nhwc_micro_ab.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c1/nhwc_micro_ab.json'))
keys=list(d)
for k in d[keys[0]]:
    print(k, [d[r][k]["ms"] for r in keys])
PY
for m in 0 1 0 1; do VS_MICRO_SCALAR=$m PYTHONPATH=. timeout 200 python tools/split_conv_micro.py r5c1/split_conv_scalar${m} 2>&1 | tr '\n' ' '; echo; done
