# the features' BatchNorm backward (8 channels, [B][T][8][F] fp32): workgroups per channel of its two passes (temporary env knobs)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
one() { n=$1; shift
  rm -rf /tmp/tr_$n
  (cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tr_$n -o t -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-extras > /tmp/tr_$n.log 2>&1)
  f=$(find /tmp/tr_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n $*"
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    for k in ('bn_act_bwd_stats_kernel','bn_act_bwd_apply_kernel','nhwc_conv_last_bwd_kernel'):
        if k in n: print('   %-40s calls %4s  avg %8.1f us' % (k, r['Calls'], float(r['AverageNs'])/1e3))
PY
  tail -1 /tmp/tr_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'])" 2>/dev/null
}
one A
one B VS_DEV_FBS=1024 VS_DEV_FBA=1024
one C VS_DEV_FBS=4096 VS_DEV_FBA=4096
one D VS_DEV_FBS=512 VS_DEV_FBA=100000
