# cnn1's backward (the step's last kernel, VALU-bound, 2048 looping workgroups with a 576-atomic flush each): workgroup count by a temporary knob
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$n', d['ms_per_step'], d['value'], 'bwd_edge', s['bwd_edge'], 'wgrad_cnn2', s['wgrad_cnn2'])"; }
for r in 1 2; do
  run g2048 VS_DEV_C1B=2048
  run g4096 VS_DEV_C1B=4096
  run g8192 VS_DEV_C1B=8192
  run g1024 VS_DEV_C1B=1024
done
