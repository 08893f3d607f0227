#!/bin/bash
set -u
O=gpurun_out/r5c10
mkdir -p $O
export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print(sys.argv[2],"ms/step",d["ms_per_step"],"cal",d["box_calibration"]["ms"],"bwd_bn",s.get("bwd_bn"),"wgrad",[s[f"wgrad_cnn{i}"] for i in range(3,8)],"dgrad",[s[f"dgrad_cnn{i}"] for i in range(3,8)])
PY
}
run def_a VOICESPLIT_BWD_APPLY_BLOCKS=0
run b1024 VOICESPLIT_BWD_APPLY_BLOCKS=1024
run b512 VOICESPLIT_BWD_APPLY_BLOCKS=512
run b256 VOICESPLIT_BWD_APPLY_BLOCKS=256
run b128 VOICESPLIT_BWD_APPLY_BLOCKS=128
run def_b VOICESPLIT_BWD_APPLY_BLOCKS=0
run serial VOICESPLIT_BWD_APPLY_BLOCKS=0
