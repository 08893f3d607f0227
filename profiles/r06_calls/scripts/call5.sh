set -u
export TMPDIR=/tmp
O=gpurun_out/r6c5
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout=1500 2>&1 | tail -15 > $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/bench_train.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c5/bench_train.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['config'].get('also_measured'))[:900])
print(json.dumps(d.get('eval_latency'))[:400])
print(json.dumps(d.get('cpu_baseline'))[:300])
PY
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --force-collectives 2>$O/forced.err | tail -1 > $O/bench_forced.json; python -c "import json;d=json.load(open('$O/bench_forced.json'));print('forced', d['ms_per_step'], json.dumps(d.get('rccl'))[:600])"; tail -3 $O/forced.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --force-collectives --loss-lag 1 2>/dev/null | tail -1 > $O/bench_forced_lag.json; python -c "import json;d=json.load(open('$O/bench_forced_lag.json'));print('forced lag1', d['ms_per_step'])"
timeout 400 python bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 4 --data files 2>$O/files.err | tail -1 > $O/bench_files.json; python -c "import json;d=json.load(open('$O/bench_files.json'));print('files', d['ms_per_step'], json.dumps(d.get('feeding'))[:300])"; tail -3 $O/files.err
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | tail -1 > $O/bench_resident.json; python -c "import json;d=json.load(open('$O/bench_resident.json'));print('resident', d['ms_per_step'])"
