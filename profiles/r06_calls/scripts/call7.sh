set -u
export TMPDIR=/tmp
O=gpurun_out/r6c7
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], 'bwd_bn', s['bwd_bn'], 'wgrad3', s['wgrad_cnn3'], 'dgrad3', s['dgrad_cnn3'], 'cnn1', s['cnn1'], 'lstm_gemm', s['lstm_gemm'])"; }
for rep in 1 2; do
timeout 300 $B 2>/dev/null | tail -1 > $O/resident_$rep.json; show $O/resident_$rep.json resident_normal
VS_EXP_SIDE_PRIO=1 timeout 300 $B 2>/dev/null | tail -1 > $O/resident_hi_$rep.json; show $O/resident_hi_$rep.json resident_high
VS_EXP_SIDE_PRIO=2 timeout 300 $B 2>/dev/null | tail -1 > $O/resident_lo_$rep.json; show $O/resident_lo_$rep.json resident_low
VS_EXP_SIDE_PRIO=1 timeout 300 $B --force-collectives 2>/dev/null | tail -1 > $O/forced_hi_$rep.json; show $O/forced_hi_$rep.json forced_high; python -c "import json;print(json.dumps(json.load(open('$O/forced_hi_$rep.json'))['rccl'].get('split_allreduce_calibration')))"
VS_EXP_SIDE_PRIO=2 timeout 300 $B --force-collectives 2>/dev/null | tail -1 > $O/forced_lo_$rep.json; show $O/forced_lo_$rep.json forced_low; python -c "import json;print(json.dumps(json.load(open('$O/forced_lo_$rep.json'))['rccl'].get('split_allreduce_calibration')))"
VS_EXP_SIDE_PRIO=2 timeout 300 $B --force-collectives --loss-lag 1 2>/dev/null | tail -1 > $O/forced_lo_lag_$rep.json; show $O/forced_lo_lag_$rep.json forced_low_lag1
done
VOICESPLIT_DETERMINISTIC=1 timeout 300 $B 2>/dev/null | tail -1 > $O/det.json; show $O/det.json deterministic
timeout 900 python -m pytest tests/test_gpu_b64_backward.py::test_deterministic_mode_reruns_bit_identically tests/test_gpu_trainer.py -q -x -s --timeout=900 2>&1 | grep -v Warning | tail -8
