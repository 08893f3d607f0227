# round 6, call 2: the one-micro-op-per-MFMA epilogue (VS_OPT_CONV_EPILOGUE = 1): unit tests under it, micro A/B, step A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r6c2
mkdir -p $O
VOICESPLIT_CONV_EPILOGUE=1 timeout 1200 python -m pytest tests/test_gpu_nhwc.py -q -x --timeout=900 2>&1 | tail -15 > $O/pytest_nhwc_fine.log; tail -4 $O/pytest_nhwc_fine.log
VS_MICRO_FINE_AB=1 VS_MICRO_WGRAD=0 timeout 600 python tools/nhwc_micro.py > $O/nhwc_micro_fine_ab.json 2>$O/micro.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c2/nhwc_micro_fine_ab.json'))
ks=list(d)
for k in d[ks[0]]:
    print(k, [d[m][k]['ms'] for m in ks])
PY
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
for rep in 1 2; do
  for ep in 0 1; do
    VOICESPLIT_CONV_EPILOGUE=$ep timeout 300 $B 2>/dev/null | tail -1 > $O/step_ep${ep}_$rep.json; python -c "import json;d=json.load(open('$O/step_ep${ep}_$rep.json'));print('ep $ep', d['ms_per_step'], d['roofline'].get('frac'), {k:d['stage_ms'][k] for k in ('cnn3','cnn7','dgrad_cnn3','dgrad_cnn7','wgrad_cnn3','bwd_bn','fwd_bn')})"
  done
done
VOICESPLIT_CONV_EPILOGUE=1 timeout 1500 python -m pytest tests/test_gpu_b64.py tests/test_gpu_b64_backward.py tests/test_gpu_bf16.py -q -x --timeout=1200 2>&1 | tail -15 > $O/pytest_b64_fine.log; tail -4 $O/pytest_b64_fine.log
