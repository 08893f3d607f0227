set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== bf16 tests"; timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q 2>&1 | tail -12 | cut -c1-600
echo "=== bench bf16"; timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --conv-math bf16 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print({k:round(v,2) for k,v in d['stage_ms'].items() if v})"
echo "=== bench default"; timeout 300 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
for f in gpurun_out/errors_bf16_*.json; do echo $f; python -c "
import json,sys
t=json.load(open('$f'))
print({k:float('%.3g'%v) for k,v in t.items() if k.startswith('fwd')})
g={k:v for k,v in t.items() if k.startswith('grad/')}; c={k:v for k,v in t.items() if k.startswith('cos/')}
print('grad max', max(g.values()), max(g,key=g.get), ' cos min', min(c.values()), min(c,key=c.get))"; done
