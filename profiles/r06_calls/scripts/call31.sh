# grid caps of the other streaming kernels of the step, by a (temporary) environment knob: per-kernel means from rocprofv3 --stats
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c31; mkdir -p $O
one() { # name, env...
  n=$1; shift
  rm -rf /tmp/tr_$n
  (cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tr_$n -o t -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-extras > /tmp/tr_$n.log 2>&1)
  f=$(find /tmp/tr_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n $*"
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    for k in ('nhwc_conv_first_kernel','nhwc_conv_last_kernel','bn_apply_feat_bf16','cvt_rows_bf16','nhwc_bn_apply_kernel','nhwc_conv_kernel<5, 5, 2'):
        if k in n: print('   %-40s calls %4s  avg %8.1f us' % (k, r['Calls'], float(r['AverageNs'])/1e3))
PY
  tail -1 /tmp/tr_$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step', d['ms_per_step'])" 2>/dev/null
}
one A
one B VS_DEV_CNN1=16384 VS_DEV_CNN8=8192 VS_DEV_FEAT=10000000 VS_DEV_CVT=10000000
one C VS_DEV_CNN1=10000000 VS_DEV_CNN8=32768 VS_DEV_FEAT=32768 VS_DEV_CVT=32768
one D VS_DEV_CNN1=65536 VS_DEV_CNN8=16384
one E VS_DEV_CNN1=4096 VS_DEV_CNN8=4096
