set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c10
mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$name -o t -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
  f=$(find $O/$name -name "*kernel_trace.csv" | head -1)
  echo "== $name: $(python -c "import json;print(json.load(open('$O/$name.json'))['ms_per_step'])" 2>/dev/null)"
  python tools/trace_window.py $f nhwc_conv_last_bwd 900 1100 | grep -v "q5"
}
run base A=1
run zero VS_EXP_ZERO=1
run prio_normal VS_EXP_PRIO_NORMAL=1
run zero_prio_normal VS_EXP_ZERO=1 VS_EXP_PRIO_NORMAL=1
B="python bench.py --no-extras --no-cpu-baseline --steps 10"
for rep in 1 2; do
for v in "A=1" "VS_EXP_ZERO=1" "VS_EXP_PRIO_NORMAL=1" "VS_EXP_ZERO=1 VS_EXP_PRIO_NORMAL=1"; do
  echo "$v: $(env $v timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['stage_ms']['bwd_edge'], d['stage_ms']['bwd_lstm_gemm'])")"
done
done
