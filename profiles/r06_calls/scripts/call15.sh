set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c15
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_audio.py tests/test_gpu_loss.py -q -x --timeout=600 2>&1 | grep -v Warning | tail -6
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2', d['ms_per_step'], d['value'], d['box_calibration']['tflops'])"; }
for rep in 1 2; do
timeout 300 $B 2>/dev/null | tail -1 > $O/resident_$rep.json; show $O/resident_$rep.json resident
timeout 400 $B --data files 2>/dev/null | tail -1 > $O/files_$rep.json; show $O/files_$rep.json files_w14
timeout 400 $B --data files --workers 4 2>/dev/null | tail -1 > $O/files4_$rep.json; show $O/files4_$rep.json files_w4
done
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_forward.py -q -x --timeout=600 2>&1 | grep -v Warning | tail -4
