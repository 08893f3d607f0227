#!/bin/bash
set -u
O=gpurun_out/r5c13
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_forward.py tests/test_gpu_trainer.py tests/test_gpu_b64.py tests/test_gpu_boundary.py -q -x --timeout=600 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
run() { n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print(sys.argv[2],"ms/step",d["ms_per_step"],"utt/s",d["value"],"lstm_gemm",s["lstm_gemm"],"head",s["head"],"cnn2",s["cnn2"],"cnn3",s["cnn3"],"frac",d["roofline"]["frac"],"gemm",d["roofline"]["lstm_input_gemm"]["frac"])
PY
}
run pro1_a VOICESPLIT_FWD_PROLOGUE=1
run pro0_a VOICESPLIT_FWD_PROLOGUE=0
run pro1_b VOICESPLIT_FWD_PROLOGUE=1
run pro0_b VOICESPLIT_FWD_PROLOGUE=0
