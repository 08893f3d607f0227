set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for rep in 1 2; do
for v in old new; do
cp $R/tools/$v.so.bin $R/voicesplit_amd/libvoicesplit_hip.so
for c in xg dfeat dW_ih; do
VS_MICRO_ONLY=$c:f16x3 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/p_$c -o t -f csv -- python $R/tools/gemm_micro.py > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob('/tmp/p_$c/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_f16x3' in r['Name']: print('$v $c', r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
rm -rf /tmp/p_$c
done
done
done
