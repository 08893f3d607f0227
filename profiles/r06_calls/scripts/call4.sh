set -u
export TMPDIR=/tmp
O=gpurun_out/r6c4
mkdir -p $O
VS_DEBUG_EP=2 python profiles/r06_calls/scripts/debug_fine.py 2>&1 | grep -v " 0 bad" | grep -v "identical to each other: True" | tail -30
VOICESPLIT_CONV_EPILOGUE=2 timeout 1200 python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_bf16.py tests/test_gpu_forward.py -q --timeout=900 2>&1 | tail -8
VS_MICRO_FINE_AB=0,1,2 VS_MICRO_WGRAD=0 timeout 900 python tools/nhwc_micro.py > $O/nhwc_micro_ab.json 2>$O/micro.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c4/nhwc_micro_ab.json'))
ks=list(d)
print(ks)
for k in d[ks[0]]:
    print(k, [d[m][k]['ms'] for m in ks])
PY
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
for rep in 1 2; do
  for ep in 0 1 2; do
    VOICESPLIT_CONV_EPILOGUE=$ep timeout 300 $B 2>/dev/null | tail -1 > $O/step_ep${ep}_$rep.json; python -c "import json;d=json.load(open('$O/step_ep${ep}_$rep.json'));print('ep $ep', d['ms_per_step'], d['roofline'].get('frac'), {k:d['stage_ms'][k] for k in ('cnn2','cnn3','cnn7','dgrad_cnn2','dgrad_cnn3','dgrad_cnn7','wgrad_cnn3','bwd_bn','fwd_bn')})"
  done
done
VOICESPLIT_CONV_EPILOGUE=2 timeout 1500 python -m pytest tests/test_gpu_b64.py tests/test_gpu_b64_backward.py -q -x --timeout=1200 2>&1 | tail -6
