# fp32-class step: workgroups per channel of the NCHW BatchNorm backward passes (on the caller's stream in front of every data gradient)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 300 python bench.py --conv-math f16x3 --no-extras --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$n', d['ms_per_step'], d['value'], 'bwd_bn', s['bwd_bn'], 'wgrad_cnn4', s['wgrad_cnn4'], 'dgrad_cnn4', s['dgrad_cnn4'])"; }
for r in 1 2; do
  run base
  run s64_a128 VS_DEV_BS=64 VS_DEV_BA=128
  run s128_a512 VS_DEV_BS=128 VS_DEV_BA=512
  run s32_a512 VS_DEV_BA=512
  run s256_a2048 VS_DEV_BS=256 VS_DEV_BA=2048
done
