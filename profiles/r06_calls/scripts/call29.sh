set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c29; mkdir -p $O
for w in 2048 4096 8192 16384 32768 65536 1000000; do
  echo "== wide cap $w"
  VS_DEV_WIDE=$w timeout 300 python tools/edge_micro.py 2>/dev/null | python -c "
import sys,json
d=json.load(sys.stdin)
for k in ('BatchNorm + mish apply','BatchNorm apply, no activation','BatchNorm backward from dy (one pass)'):
    print('  %-45s %.3f ms  %.2f TB/s' % (k, d[k]['ms'], d[k]['TB/s']))
"
done 2>&1 | tee $O/wide.txt
