set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$n', d['ms_per_step'], d['value'], 'bwd_bn', s['bwd_bn'], 'bwd_edge', s['bwd_edge'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'])"; }
for r in 1 2 3; do
  run base
  run feat_bwd_grids VS_DEV_FBS=512 VS_DEV_FBA=4096
done
