#!/bin/bash
for abl in 0 2 4 6 14; do
  echo "== VOICESPLIT_GEMM_ABL=$abl"
  VOICESPLIT_GEMM_ABL=$abl VS_MICRO_ONLY=bf16 timeout 100 python tools/gemm_micro.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('   ', {k.split()[0]: v['ms'] for k,v in d.items()})
"
done
