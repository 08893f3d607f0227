set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 256 16; do
timeout 300 python bench.py --mode forward --batch $b --no-cpu-baseline --steps 5 --warmup 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('forward B=$b', d['value'], d['ms_per_step'])"
done
for b in 16 2; do
timeout 300 python bench.py --batch $b --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('train B=$b', d['value'], d['ms_per_step'])"
done
