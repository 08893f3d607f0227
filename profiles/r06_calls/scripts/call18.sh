set -u
R=r06
O=gpurun_out/final
mkdir -p $O
bash tools/profile_gpu.sh ${R}_forward --mode forward 2>&1 | tail -2
bash tools/profile_gpu.sh ${R}_train_bf16 --conv-math bf16 2>&1 | tail -2
F=$(find gpurun_out/prof_${R}_train_bf16/trace -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python tools/step_timeline.py $F > $O/${R}_step_timeline.txt && head -3 $O/${R}_step_timeline.txt
[ -n "$F" ] && python tools/dispatch_census.py $F --steps 10 > $O/${R}_dispatch_census.txt && head -2 $O/${R}_dispatch_census.txt
