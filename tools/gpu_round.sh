set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -q -x 2>&1 | tail -2 | cut -c1-300
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -f csv -- python $R/bench.py --mode forward --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob('/tmp/pp/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('gemm_pre','split_rows')): print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
timeout 200 rocprofv3 --kernel-trace --kernel-include-regex "gemm_pre" --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pq -o p -f csv -- python $R/bench.py --mode forward --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pq.log 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Counter_Name']].append(float(r['Counter_Value'])); d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
t={k:sum(v)/len(v) for k,v in acc.items()}
g=t['GRBM_GUI_ACTIVE']/8
print('dur', d, 'mfma busy', t['SQ_VALU_MFMA_BUSY_CYCLES']/(g*1024), 'clk', g/d/1e6, 'wait_any', t['SQ_WAIT_ANY']/t['SQ_WAVE_CYCLES'], 'lds conf', t['SQ_LDS_BANK_CONFLICT']/t['SQ_LDS_IDX_ACTIVE'])
PY
