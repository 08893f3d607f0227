set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6c16
mkdir -p $O
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], d['value'], 'cal', d['box_calibration']['tflops'], 'cnn3', s.get('cnn3'), 'cnn2', s.get('cnn2'), 'lstm_gemm', s.get('lstm_gemm'), 'frac', d['roofline']['frac'])"; }
for rep in 1 2 3; do
(cd build/r5tree && timeout 300 python bench.py --mode forward --no-cpu-baseline 2>/dev/null | tail -1 > $O/r5_forward_$rep.json); show $O/r5_forward_$rep.json r5_forward
timeout 300 python bench.py --mode forward --no-cpu-baseline 2>/dev/null | tail -1 > $O/r6_forward_$rep.json; show $O/r6_forward_$rep.json r6_forward
(cd build/r5tree && timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > $O/r5_train_$rep.json); show $O/r5_train_$rep.json r5_train
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 2>/dev/null | tail -1 > $O/r6_train_$rep.json; show $O/r6_train_$rep.json r6_train
done
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_forward.py -q -x --timeout=600 2>&1 | grep -E "passed|failed|error" | tail -3
