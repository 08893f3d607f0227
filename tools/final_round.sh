# Round-end GPU script (one gpurun call): full -m gpu suite, smoke(), the bench lines, the probes, the rocprofv3 profile.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/final_round.sh r06'
# Results land in gpurun_out/final/; copy what is to be kept into profiles/ (tools/collect_final.sh <round>).
set -u
R=${1:-r06}
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/${R}_pytest_gpu.log; cat $O/${R}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
T0=$(date +%s)
timeout 900 python bench.py 2> $O/bench_train.err | tail -1 > $O/${R}_bench_train.json; cut -c1-300 $O/${R}_bench_train.json
echo "default bench.py run: $(( $(date +%s) - T0 )) s wall" | tee $O/${R}_bench_train_wall.txt
timeout 300 python bench.py --mode forward --no-cpu-baseline 2> $O/bench_forward.err | tail -1 > $O/${R}_bench_forward.json; cut -c1-200 $O/${R}_bench_forward.json
timeout 300 python bench.py --mode forward --conv-math bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_forward_bf16.json; cut -c1-200 $O/${R}_bench_forward_bf16.json
timeout 300 python bench.py --mode longform --conv-math bf16 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_longform_bf16.json; cut -c1-200 $O/${R}_bench_longform_bf16.json
timeout 300 python bench.py --mode longform --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_longform.json; cut -c1-200 $O/${R}_bench_longform.json
timeout 300 python bench.py --serial-backward --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_serial_backward.json; cut -c1-200 $O/${R}_bench_train_serial_backward.json
timeout 300 python bench.py --model voicefilter --loss powerlaw --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_voicefilter_powerlaw.json; cut -c1-200 $O/${R}_bench_train_voicefilter_powerlaw.json
timeout 200 python bench.py --batch 2 --steps 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_b2.json; cut -c1-200 $O/${R}_bench_train_b2.json
# [r6] the same step four more ways: 100 steps, the N > 1 code path on one rank, fed from files, deterministic mode
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_100steps.json; cut -c1-200 $O/${R}_bench_train_100steps.json
timeout 300 python bench.py --force-collectives --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_forced_collectives.json; cut -c1-200 $O/${R}_bench_train_forced_collectives.json
timeout 400 python bench.py --data files --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_data_files.json; cut -c1-200 $O/${R}_bench_train_data_files.json
VOICESPLIT_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_deterministic.json; cut -c1-200 $O/${R}_bench_train_deterministic.json
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/${R}_bench_train_again.json; cut -c1-200 $O/${R}_bench_train_again.json
timeout 300 python tools/nhwc_micro.py > $O/${R}_nhwc_micro.json 2>/dev/null
timeout 200 python tools/edge_micro.py > $O/${R}_edge_micro.json 2>/dev/null
timeout 200 python tools/gemm_epilogue_probe.py > $O/${R}_gemm_epilogue_probe.json 2>/dev/null
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_train_torchrun_n1.json; cut -c1-160 $O/${R}_bench_train_torchrun_n1.json
VS_MICRO_ONLY=bf16 timeout 300 python tools/gemm_micro.py > $O/${R}_gemm_micro.json 2>/dev/null
{ for p in gemm_issue_probe occupancy_probe epilogue_slot_probe stream_probe; do echo "== tools/$p"; timeout 120 tools/$p; done; } > $O/${R}_probes.txt 2>&1
{ timeout 200 python tools/lstm_time.py 64; timeout 200 python tools/lstm_time.py 2; } 2>/dev/null | grep "B=" > $O/${R}_lstm_time.txt
[ -f voicesplit_amd/libvoicesplit_hip_abl.so ] && timeout 300 python tools/wgrad_ablation.py > $O/${R}_wgrad_ablation.json 2>/dev/null
PYTHONPATH=. timeout 200 python tools/split_conv_micro.py final/${R}_split_conv_micro > /dev/null 2>&1
bash tools/profile_gpu.sh ${R}_forward --mode forward 2>&1 | tail -2
bash tools/profile_gpu.sh ${R}_train_bf16 --conv-math bf16 2>&1 | tail -2
# [r6] counters for the other legs' roofline.traffic: bf16 forward, long-form (256 windows per batch) in both arithmetics, the fp32-class step
bash tools/profile_gpu.sh ${R}_forward_bf16 --mode forward --conv-math bf16 2>&1 | tail -1
bash tools/profile_gpu.sh ${R}_longform --mode longform 2>&1 | tail -1
bash tools/profile_gpu.sh ${R}_longform_bf16 --mode longform --conv-math bf16 2>&1 | tail -1
bash tools/profile_gpu.sh ${R}_train_f16x3 --conv-math f16x3 2>&1 | tail -1
# device idle gaps of one training step, from the kernel trace of the profile run
F=$(find gpurun_out/prof_${R}_train_bf16/trace -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python tools/step_timeline.py $F > $O/${R}_step_timeline.txt && head -3 $O/${R}_step_timeline.txt
[ -n "$F" ] && python tools/dispatch_census.py $F --steps 10 > $O/${R}_dispatch_census.txt && head -2 $O/${R}_dispatch_census.txt
