set -u
export TMPDIR=/tmp
O=gpurun_out/r6c6
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], 'bwd_bn', s['bwd_bn'], 'wgrad3', s['wgrad_cnn3'], 'dgrad3', s['dgrad_cnn3'])"; }
timeout 300 $B 2>/dev/null | tail -1 > $O/resident.json; show $O/resident.json resident
timeout 300 $B --force-collectives --split-allreduce off 2>/dev/null | tail -1 > $O/forced.json; show $O/forced.json forced
VS_BENCH_INIT_GROUP_ONLY=1 timeout 300 $B 2>/dev/null | tail -1 > $O/group_only.json; show $O/group_only.json group_only
GPU_MAX_HW_QUEUES=8 timeout 300 $B --force-collectives --split-allreduce off 2>/dev/null | tail -1 > $O/forced_q8.json; show $O/forced_q8.json forced_q8
VS_EXP_SIDE_PRIO=1 timeout 300 $B --force-collectives --split-allreduce off 2>/dev/null | tail -1 > $O/forced_hi.json; show $O/forced_hi.json forced_side_high
VS_EXP_SIDE_PRIO=2 timeout 300 $B --force-collectives --split-allreduce off 2>/dev/null | tail -1 > $O/forced_lo.json; show $O/forced_lo.json forced_side_low
VS_EXP_SIDE_PRIO=1 timeout 300 $B 2>/dev/null | tail -1 > $O/resident_hi.json; show $O/resident_hi.json resident_side_high
timeout 900 python -m pytest tests/test_gpu_b64_backward.py::test_deterministic_mode_reruns_bit_identically tests/test_gpu_trainer.py::test_bf16_trains_on_real_audio_at_the_references_hyper_parameters tests/test_gpu_bf16.py::test_every_schedule_of_the_bf16_step_gives_the_same_bits -q -x -s --timeout=900 2>&1 | grep -v Warning | tail -12
python - <<'PY'
import json
t=json.load(open('gpurun_out/trajectory_real_audio.json'))
for k,v in t.items(): print(k, [round(x,2) for x in v[:5]], [round(x,2) for x in v[-3:]], 'min', round(min(v),2))
PY
VOICESPLIT_DETERMINISTIC=1 timeout 300 $B 2>/dev/null | tail -1 > $O/det.json; show $O/det.json deterministic
timeout 300 $B 2>/dev/null | tail -1 > $O/resident2.json; show $O/resident2.json resident
