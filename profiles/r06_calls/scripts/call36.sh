# gradient bucket views on 16-byte boundaries: dW_ih then takes the interleaved LDS-DMA kernel instead of the round-3 one (vec_ok)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { n=$1; shift; env "$@" timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$n', d['ms_per_step'], d['value'], 'bwd_edge', s['bwd_edge'], 'bwd_lstm_gemm', s['bwd_lstm_gemm'], 'bwd_bn', s['bwd_bn'])"; }
for r in 1 2 3; do
  run unaligned VS_DEV_BUCKET_ALIGN=0
  run aligned VS_DEV_BUCKET_ALIGN=1
done
(cd /tmp && VS_DEV_BUCKET_ALIGN=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 9 --warmup 1 --no-cpu-baseline --no-extras > /tmp/tr.log 2>&1)
grep "gemm_bf16" $(find /tmp/tr -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-160
