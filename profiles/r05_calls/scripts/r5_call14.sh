#!/bin/bash
set -u
O=gpurun_out/r5c14
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_loss.py tests/test_gpu_audio.py tests/test_gpu_trainer.py tests/test_gpu_bf16.py -q -x --timeout=600 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
run() { n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print(sys.argv[2],"ms/step",d["ms_per_step"],"utt/s",d["value"])
PY
}
run f16x3_a VOICESPLIT_LOSS_GEMM_FP32=0
run fp32_a VOICESPLIT_LOSS_GEMM_FP32=1
run f16x3_b VOICESPLIT_LOSS_GEMM_FP32=0
run fp32_b VOICESPLIT_LOSS_GEMM_FP32=1
