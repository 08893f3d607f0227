set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['stage_ms']); print(sum(v for v in d['stage_ms'].values() if v))"
