set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c19
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], d['value'], 'bwd_bn', s['bwd_bn'], 'wgrad3', s['wgrad_cnn3'], 'dgrad3', s['dgrad_cnn3'], 'wgrad2', s['wgrad_cnn2'], 'edge', s['bwd_edge'])"; }
for rep in 1 2 3; do
VS_EXP_SCHED=0 timeout 300 $B 2>/dev/null | tail -1 > $O/old_$rep.json; show $O/old_$rep.json old_roles
VS_EXP_SCHED=1 timeout 300 $B 2>/dev/null | tail -1 > $O/new_$rep.json; show $O/new_$rep.json new_roles
done
timeout 1500 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py tests/test_gpu_trainer.py tests/test_gpu_backward.py -q -x --timeout=900 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/trace_bench.json 2> $O/trace_bench.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py $f "nhwc_wgrad_kernel<7, 1" 9000 1500 | grep -v " q5 " | tail -60
