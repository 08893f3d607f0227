#!/bin/bash
# timelines of one step with the head's data gradients on gemm_bf16 (1) and on the generic kernel (0)
mkdir -p gpurun_out/r5c23
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf /tmp/tl$mode
  VOICESPLIT_HEAD_BWD_GEMM=$mode rocprofv3 --kernel-trace -d /tmp/tl$mode -o trace -f csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2>&1
  F=$(find /tmp/tl$mode -name "*kernel_trace.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/step_timeline.py $F > $GRAFT_REPO_ROOT/gpurun_out/r5c23/timeline_head_bwd_gemm$mode.txt
  head -1 $GRAFT_REPO_ROOT/gpurun_out/r5c23/timeline_head_bwd_gemm$mode.txt
done
