#!/bin/bash
# One GPU lease, parameterised (replaces the per-call scratch scripts of round 3):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <name> "<pytest args or ->" [bench args... | -]'
# runs, in order: pytest (when its argument is not "-"), bench.py with the remaining arguments (when not "-"); everything
# lands in gpurun_out/<name>/.
set -u
N=${1:?name}; shift
PT=${1:--}; shift || true
O=gpurun_out/$N
mkdir -p $O
export TMPDIR=/tmp
if [ "$PT" != "-" ]; then
  timeout 1500 python -m pytest $PT -q -x --timeout=900 2>&1 | tail -40 > $O/pytest.log
  tail -15 $O/pytest.log
fi
if [ $# -gt 0 ] && [ "$1" != "-" ]; then
  timeout 900 python bench.py "$@" 2> $O/bench.err | tail -1 > $O/bench.json
  cut -c1-400 $O/bench.json
  tail -3 $O/bench.err
fi
