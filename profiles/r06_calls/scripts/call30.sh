# one-sweep grids for the BatchNorm apply pass and the stand-alone BatchNorm backward pass: tests, then two binaries alternating
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6c30; mkdir -p $O
cp voicesplit_amd/libvs_new.so voicesplit_amd/libvoicesplit_hip.so
timeout 1500 python -m pytest -q -x --timeout=900 tests/test_gpu_nhwc.py tests/test_gpu_b64.py tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py 2>&1 | tail -4
run() { timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['stage_ms']['fwd_bn'], d['stage_ms']['bwd_bn'], d['roofline']['frac'])"; }
for r in 1 2 3; do
  cp voicesplit_amd/libvs_base.so voicesplit_amd/libvoicesplit_hip.so; run base
  cp voicesplit_amd/libvs_new.so voicesplit_amd/libvoicesplit_hip.so; run one_sweep
done 2>&1 | tee $O/ab.txt
