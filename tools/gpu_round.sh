set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py -m gpu -q -x -k "conv64 or dgrad" 2>&1 | tail -2 | cut -c1-300
for rep in 1 2; do for v in old new; do
cp $R/tools/$v.so.bin $R/voicesplit_amd/libvoicesplit_hip.so
VS_MICRO_ONLY=7x1 VS_MICRO_REPS=5 timeout 200 python tools/conv_micro.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', {k:v['ms'] for k,v in d.items() if isinstance(v,dict)})"
done; done
