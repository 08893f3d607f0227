#!/bin/bash
# kernel timeline of one bf16 training step: where the device idles
O=$PWD/gpurun_out/r3c31
mkdir -p $O
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace -f csv -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
echo "trace rc=$?"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $REPO/tools/step_timeline.py $F > $O/timeline_b64.txt; head -32 $O/timeline_b64.txt
timeout 600 rocprofv3 --kernel-trace -d $O/trace2 -o trace -f csv -- python $REPO/bench.py --batch 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/trace2.log 2>&1
F=$(find $O/trace2 -name "*kernel_trace.csv" | head -1)
python $REPO/tools/step_timeline.py $F > $O/timeline_b2.txt; head -8 $O/timeline_b2.txt
find $O -type f ! -name "*.txt" ! -name "*.log" -delete
