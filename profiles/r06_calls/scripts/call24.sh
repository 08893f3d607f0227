# counters for the legs whose roofline.traffic was null: bf16 forward, long-form in both arithmetics, fp32-class step
set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_gpu.sh r06_forward_bf16 --mode forward --conv-math bf16 2>&1 | tail -3
bash tools/profile_gpu.sh r06_longform --mode longform 2>&1 | tail -3
bash tools/profile_gpu.sh r06_longform_bf16 --mode longform --conv-math bf16 2>&1 | tail -3
bash tools/profile_gpu.sh r06_train_f16x3 --conv-math f16x3 2>&1 | tail -3
du -sh gpurun_out/prof_r06_*
