set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c22
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
cp voicesplit_amd/libvoicesplit_hip.so /tmp/new.so
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], d['value'], 'cnn1', s['cnn1'])"; }
for rep in 1 2 3; do
cp voicesplit_amd/libvoicesplit_hip_prev.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/prev_$rep.json; show $O/prev_$rep.json prev
cp /tmp/new.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/new_$rep.json; show $O/new_$rep.json new
done
VOICESPLIT_DETERMINISTIC=1 timeout 300 $B 2>/dev/null | tail -1 > $O/det.json; show $O/det.json deterministic
timeout 1800 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_b64.py tests/test_gpu_bf16.py tests/test_gpu_nhwc.py tests/test_gpu_forward.py tests/test_gpu_trainer.py -q -x --timeout=900 2>&1 | grep -E "passed|failed|error" | tail -3
