set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c23
mkdir -p $O
timeout 900 python -m pytest -q -x --timeout=600 tests/test_gpu_loss.py "tests/test_gpu_bf16.py::test_forward_in_batch_parts_gives_the_whole_batch_step" "tests/test_gpu_bf16.py::test_every_schedule_of_the_bf16_step_gives_the_same_bits" tests/test_gpu_trainer.py 2>&1 | tail -15
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d['value'], d['roofline']['frac'])"
}
for r in 1 2 3; do
  run base VOICESPLIT_OVERLAP_TARGET=0 VOICESPLIT_FWD_PARTS=1
  run target VOICESPLIT_OVERLAP_TARGET=1 VOICESPLIT_FWD_PARTS=1
  run parts2 VOICESPLIT_OVERLAP_TARGET=0 VOICESPLIT_FWD_PARTS=2
  run parts4 VOICESPLIT_OVERLAP_TARGET=0 VOICESPLIT_FWD_PARTS=4
  run both2 VOICESPLIT_OVERLAP_TARGET=1 VOICESPLIT_FWD_PARTS=2
done 2>&1 | tee $O/ab.txt
