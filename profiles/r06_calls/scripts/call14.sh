set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_audio.py tests/test_gpu_forward.py -q -x --timeout=600 2>&1 | grep -v Warning | tail -6
B="python bench.py --no-extras --no-cpu-baseline --steps 20"
cp voicesplit_amd/libvoicesplit_hip.so /tmp/new.so
show() { python -c "import json,sys;d=json.load(open('$1'));s=d['stage_ms'];print('$2', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3; do
cp voicesplit_amd/libvoicesplit_hip_prev.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/prev_$rep.json; show $O/prev_$rep.json prev
cp /tmp/new.so voicesplit_amd/libvoicesplit_hip.so
timeout 300 $B 2>/dev/null | tail -1 > $O/new_$rep.json; show $O/new_$rep.json new
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/trace_bench.json 2> $O/trace_bench.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py $f sisnr_moments_kernel 700 300 | grep -v " q5 "
