#!/bin/bash
set -u
O=gpurun_out/r5c17
mkdir -p $O
export TMPDIR=/tmp
VOICESPLIT_HEAD_LEAF_SIDE=1 timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_backward.py tests/test_gpu_trainer.py -q -x --timeout=600 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
run() { n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print(sys.argv[2],"ms/step",d["ms_per_step"],"utt/s",d["value"],"bwd_head",s["bwd_head"],"bwd_lstm_rec",s["bwd_lstm_rec"])
PY
}
run side1_a VOICESPLIT_HEAD_LEAF_SIDE=1
run side0_a VOICESPLIT_HEAD_LEAF_SIDE=0
run side1_b VOICESPLIT_HEAD_LEAF_SIDE=1
run side0_b VOICESPLIT_HEAD_LEAF_SIDE=0
