set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== conv_bench"; timeout 300 tools/conv_bench 64 10 prof2 2>&1 | grep -v "^  " | tail -20
echo "=== pytest"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | cut -c1-400
echo "=== bench"; timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print({k:round(v,2) for k,v in d['stage_ms'].items() if v})"
