set -u
export TMPDIR=/tmp
O=gpurun_out/r6c8
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], 'bwd_bn', s['bwd_bn'], 'cnn1', s['cnn1'], 'cnn8', s['cnn8'], 'bwd_edge', s['bwd_edge'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'])"; }
for rep in 1 2; do
timeout 300 $B 2>/dev/null | tail -1 > $O/resident_$rep.json; show $O/resident_$rep.json resident
VOICESPLIT_DETERMINISTIC=1 timeout 300 $B 2>/dev/null | tail -1 > $O/det_$rep.json; show $O/det_$rep.json deterministic
timeout 300 $B --force-collectives 2>/dev/null | tail -1 > $O/forced_$rep.json; show $O/forced_$rep.json forced
done
timeout 1500 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py tests/test_gpu_trainer.py -q -x --timeout=900 2>&1 | grep -v Warning | tail -8
