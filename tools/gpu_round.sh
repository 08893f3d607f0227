set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "conv_edge or module_backward or golden or layerwise" 2>&1 | tail -2 | cut -c1-300
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -f csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('conv_last','conv_first','reduce_partials')): print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
