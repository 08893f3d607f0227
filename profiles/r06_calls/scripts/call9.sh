set -u
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c9
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
echo "trace: $f"; wc -l $f
python tools/dispatch_census.py $f --steps 4 | head -60
