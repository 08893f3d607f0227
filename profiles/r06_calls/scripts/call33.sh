# fp32-class step: workgroups per channel of the NCHW BatchNorm apply pass (temporary env knob), per-kernel means from rocprofv3 --stats
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6c33
one() { n=$1; shift
  rm -rf /tmp/tr_$n
  (cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tr_$n -o t -f csv -- python $GRAFT_REPO_ROOT/bench.py --conv-math f16x3 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > /tmp/tr_$n.log 2>&1)
  f=$(find /tmp/tr_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n $*"
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    for k in ('bn_apply_kernel','conv64_f16x3_pk_kernel<2, 2, 0, 0, 3, true'):
        if k in n and 'nhwc' not in n: print('   %-40s calls %4s  avg %8.1f us' % (k, r['Calls'], float(r['AverageNs'])/1e3))
PY
  tail -1 /tmp/tr_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'], 'fwd_bn', d['stage_ms']['fwd_bn'])" 2>/dev/null
}
one A
one B VS_DEV_NCHW=128
one C VS_DEV_NCHW=512
one D VS_DEV_NCHW=100000
