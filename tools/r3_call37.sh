#!/bin/bash
O=gpurun_out/r3c37
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_nhwc.py -q -k "gemm_bf16" > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
for dm in 200 000 111 222; do
  echo "== VOICESPLIT_GEMM_DMA=$dm"
  VOICESPLIT_GEMM_DMA=$dm VS_MICRO_ONLY=bf16 timeout 200 python tools/gemm_micro.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('   ',k,v)
"
done > $O/gemm_dm.txt 2>&1
cat $O/gemm_dm.txt
