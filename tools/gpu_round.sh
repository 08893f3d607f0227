set -u
export TMPDIR=/tmp
REPO=$PWD
echo "=== wgrad tests"; timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "wgrad or dgrad_and" 2>&1 | grep -E "passed|failed|error" | cut -c1-200
for M in 0 1; do for P in 0 1; do echo -n "map $M pace $P"; VS_WGRAD_MAP=$M VS_WGRAD_PACE=$P timeout 120 tools/conv_bench 64 5 wgrad 3 2>&1 | grep "selected" | cut -c40-90; done; done
mkdir -p gpurun_out/pace
cd /tmp
for M in 0 1; do for P in 0 1; do
  VS_WGRAD_MAP=$M VS_WGRAD_PACE=$P timeout 150 rocprofv3 --kernel-trace --kernel-include-regex "ring4" --pmc FETCH_SIZE -d $REPO/gpurun_out/pace/f$M$P -o pmc -f csv -- $REPO/tools/conv_bench 64 2 wgrad 3 > $REPO/gpurun_out/pace/f$M$P.log 2>&1
done; done
cd $REPO
python - <<'PY'
import csv,collections
for M in (0,1):
  for P in (0,1):
    d=collections.defaultdict(float)
    for r in csv.DictReader(open(f'gpurun_out/pace/f{M}{P}/pmc_counter_collection.csv')):
        if 'ring4' in r['Kernel_Name']: d[r['Dispatch_Id']]+=float(r['Counter_Value'])
    v=list(d.values()); print('map',M,'pace',P,'FETCH_SIZE x2 GB per launch', [round(x*2*1024/1e9,2) for x in v])
PY
for M in 0 1 0 1; do
  VS_WGRAD_MAP=$M VS_WGRAD_PACE=$M timeout 300 python bench.py --no-cpu-baseline --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['stage_ms']; print('map+pace $M', d['value'], d['ms_per_step'], 'wgrad', s['wgrad_cnn4'], s['wgrad_cnn7'])"
done
find gpurun_out/pace -type f ! -name "*.csv" ! -name "*.log" -delete
