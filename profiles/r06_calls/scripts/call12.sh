set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c12
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
cp voicesplit_amd/libvoicesplit_hip.so /tmp/new.so
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'], 'bwd_lstm_rec', s['bwd_lstm_rec'], 'bwd_bn', s['bwd_bn'], 'wgrad5', s['wgrad_cnn5'], 'dgrad5', s['dgrad_cnn5'])"; }
for rep in 1 2 3; do
cp voicesplit_amd/libvoicesplit_hip_prev.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/prev_$rep.json; show $O/prev_$rep.json prev
cp /tmp/new.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/new_$rep.json; show $O/new_$rep.json new
VS_EXP_BWD_FUSED_FOLD=1 timeout 300 $B 2>/dev/null | tail -1 > $O/fused_$rep.json; show $O/fused_$rep.json new_fused_fold
done
timeout 1500 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py -q -x --timeout=900 2>&1 | grep -v Warning | tail -4
VS_EXP_BWD_FUSED_FOLD=1 timeout 1500 python -m pytest tests/test_gpu_b64_backward.py -q -x --timeout=900 2>&1 | grep -v Warning | tail -4
