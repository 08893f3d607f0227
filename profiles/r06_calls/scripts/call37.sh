# fp32-class forward (configs[1]): 8192 workgroups for its cnn1 / cnn8 kernels (the split-plane twins of the bf16 ones), two binaries alternating
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { timeout 300 python bench.py --mode forward --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', d['ms_per_step'], d['value'], 'cnn1', s['cnn1'], 'cnn8', s['cnn8'])"; }
for r in 1 2 3; do
  cp voicesplit_amd/libvs_base.so voicesplit_amd/libvoicesplit_hip.so; run base
  cp voicesplit_amd/libvs_new.so voicesplit_amd/libvoicesplit_hip.so; run caps8192
done
