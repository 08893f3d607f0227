set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c21
mkdir -p $O
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
cp voicesplit_amd/libvoicesplit_hip.so /tmp/new.so
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], d['value'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'], 'lstm_gemm', s['lstm_gemm'])"; }
for rep in 1 2 3; do
cp voicesplit_amd/libvoicesplit_hip_prev.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/prev_$rep.json; show $O/prev_$rep.json prev
cp /tmp/new.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/new_$rep.json; show $O/new_$rep.json new
done
timeout 1800 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py tests/test_gpu_boundary.py tests/test_gpu_head.py tests/test_gpu_b64.py -q -x --timeout=900 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/trace_bench.json 2> $O/trace_bench.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py $f lstm16_bwd_persistent 300 2400 | grep -v " q5 " | grep -i "cvt\|gemm_bf16\|lstm16"
