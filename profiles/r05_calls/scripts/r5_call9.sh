#!/bin/bash
set -u
O=gpurun_out/r5c9
mkdir -p $O
export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print(sys.argv[2],"ms/step",d["ms_per_step"],"cal",d["box_calibration"]["ms"],"fwd_bn",s.get("fwd_bn"),"bwd_bn",s.get("bwd_bn"),"wgrad",[s[f"wgrad_cnn{i}"] for i in range(3,8)],"dgrad",[s[f"dgrad_cnn{i}"] for i in range(3,8)])
PY
}
run fused1_a VOICESPLIT_BN_FUSED_FINALIZE=1
run fused0_a VOICESPLIT_BN_FUSED_FINALIZE=0
run fused1_hi VOICESPLIT_BN_FUSED_FINALIZE=1 VOICESPLIT_SIDE_PRIO=1
run fused0_hi VOICESPLIT_BN_FUSED_FINALIZE=0 VOICESPLIT_SIDE_PRIO=1
run fused1_lo VOICESPLIT_BN_FUSED_FINALIZE=1 VOICESPLIT_SIDE_PRIO=2
run fused1_b VOICESPLIT_BN_FUSED_FINALIZE=1
run fused0_b VOICESPLIT_BN_FUSED_FINALIZE=0
