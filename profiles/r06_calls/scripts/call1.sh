# round 6, call 1: baseline of the round-5 binary + the new parity tests + the epilogue-slot probe + the two-pass BatchNorm A/B
set -u
export TMPDIR=/tmp
O=gpurun_out/r6c1
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_b64_backward.py tests/test_gpu_boundary.py -q -x --timeout=1200 2>&1 | tail -30 > $O/pytest.log
tail -15 $O/pytest.log
timeout 120 tools/epilogue_slot_probe > $O/epilogue_slot_probe.txt 2>&1; cat $O/epilogue_slot_probe.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
for rep in 1 2; do
  timeout 300 $B 2>/dev/null | tail -1 > $O/base_$rep.json; python -c "import json;d=json.load(open('$O/base_$rep.json'));print('base', d['ms_per_step'], d['roofline'].get('frac'))"
  for nb in 256 512 1024; do
    VOICESPLIT_BWD_DY=0 VOICESPLIT_BWD_APPLY_BLOCKS=$nb timeout 300 $B 2>/dev/null | tail -1 > $O/twopass_${nb}_$rep.json; python -c "import json;d=json.load(open('$O/twopass_${nb}_$rep.json'));print('twopass $nb', d['ms_per_step'])"
  done
done
