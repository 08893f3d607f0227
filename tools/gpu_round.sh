set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -k "full_size_layerwise or golden_gradients or fp64_oracle" 2>&1 | tail -40 > gpurun_out/t_bwd.log
timeout 300 python tools/train_step_timing.py --steps 2 --warmup 1 > gpurun_out/train_timing.log 2>&1
cat gpurun_out/t_bwd.log | tail -30; cat gpurun_out/train_timing.log; cat gpurun_out/errors_layerwise_full.json
