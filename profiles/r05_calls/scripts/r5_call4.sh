#!/bin/bash
set -u
O=gpurun_out/r5c4
mkdir -p $O
export TMPDIR=/tmp
VOICESPLIT_CONV8=3 timeout 900 python -m pytest tests/test_gpu_nhwc.py -q -x --timeout=600 -k "nhwc_conv_matches or data_gradient_weights or fused_statistics or one_hot_indexing or dy_epilogue or conv_stack_stage" 2>&1 | tail -25
VS_MICRO_CONV8_AB=1 VS_MICRO_WGRAD=0 timeout 600 python tools/nhwc_micro.py > $O/nhwc_micro_conv8.json 2> $O/nhwc_micro_conv8.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c4/nhwc_micro_conv8.json'))
keys=list(d)
for k in d[keys[0]]:
    if '5x5' in k: print(k, [d[r][k]["ms"] for r in keys])
PY
tail -3 $O/nhwc_micro_conv8.err
