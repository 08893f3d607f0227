#!/bin/bash
set -u
O=gpurun_out/r5c5
mkdir -p $O
export TMPDIR=/tmp
VS_MICRO_LIB=libvoicesplit_hip_probe.so VS_MICRO_CONV8_PROBE=1 timeout 300 python tools/nhwc_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/conv8_probe.txt
for lib in libvoicesplit_hip.so libvoicesplit_hip_pd4.so libvoicesplit_hip_pd6.so; do
  VS_MICRO_LIB=$lib VS_MICRO_CONV8_AB=1 VS_MICRO_WGRAD=0 VS_MICRO_DY=0 timeout 600 python tools/nhwc_micro.py > $O/micro_$lib.json 2> /dev/null
  python - $O/micro_$lib.json $lib <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
keys=list(d)
print(sys.argv[2], keys)
for k in d[keys[0]]:
    if '5x5' in k and ('dil1 ' in k or 'dil4 ' in k): print(k, [d[r][k]["ms"] for r in keys])
PY
done
