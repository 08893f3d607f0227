set -u
export TMPDIR=/tmp
O=gpurun_out/r6c3
mkdir -p $O
python profiles/r06_calls/scripts/debug_fine.py 2>&1 | grep -v " 0 bad" | tail -30
VOICESPLIT_CONV_EPILOGUE=1 timeout 1200 python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_bf16.py tests/test_gpu_forward.py -q --timeout=900 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_trainer.py -q -x --timeout=900 2>&1 | tail -12
