# (1) tests on the new grid caps (cnn1 / cnn8 forward 8192, feature apply and row conversion one sweep); (2) the throttled BatchNorm backward
# pass beside the weight gradient: contiguous part per workgroup instead of a strided loop, and its workgroup count (temporary env knobs)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c32; mkdir -p $O
timeout 1500 python -m pytest -q -x --timeout=900 tests/test_gpu_nhwc.py tests/test_gpu_b64.py tests/test_gpu_bf16.py tests/test_gpu_kernels.py tests/test_gpu_forward.py 2>&1 | tail -3
run() { n=$1; shift; env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$n', d['ms_per_step'], d['value'], 'bwd_bn', s['bwd_bn'], 'wgrad3-6', s['wgrad_cnn3'], s['wgrad_cnn4'], s['wgrad_cnn5'], s['wgrad_cnn6'])"; }
for r in 1 2; do
  run stride256 VS_DEV_BWDCHUNK=0
  run chunk256 VS_DEV_BWDCHUNK=1
  run chunk512 VS_DEV_BWDCHUNK=1 VS_DEV_BWDGRID=512
  run chunk1024 VS_DEV_BWDCHUNK=1 VS_DEV_BWDGRID=1024
  run stride512 VS_DEV_BWDCHUNK=0 VS_DEV_BWDGRID=512
done 2>&1 | tee $O/ab.txt
