#!/bin/bash
set -u
O=gpurun_out/r5c3
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_b64.py -q --timeout=900 2>&1 | tail -5
for p in 0 1 2 3 0 1; do
  VOICESPLIT_MFMA_PRIO=$p timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 2> $O/bench_prio$p.err | tail -1 > $O/bench_prio$p.json
  python - $O/bench_prio$p.json $p <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d["stage_ms"]
print("prio",sys.argv[2],"ms/step",d["ms_per_step"],"cal",d["box_calibration"]["ms"],"wgrad3-7",[s[f"wgrad_cnn{i}"] for i in range(3,8)],"dgrad",[s[f"dgrad_cnn{i}"] for i in range(3,8)],"bwd_bn",s.get("bwd_bn"),"fwd_bn",s.get("fwd_bn"))
PY
done
timeout 600 python bench.py --no-cpu-baseline 2> $O/bench_full.err | tail -1 > $O/bench_full.json; cut -c1-300 $O/bench_full.json; tail -3 $O/bench_full.err
