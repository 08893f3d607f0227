set -u
export TMPDIR=/tmp
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_trainer.py tests/test_gpu_bf16.py -m gpu -q -x -k "side_stream or golden or training_step or module or trainer or bf16 or full" 2>&1 | grep -E "passed|failed|error|Error" | cut -c1-300
echo "=== bench overlap"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('stage_ms'))"
echo "=== bench B=2"; timeout 600 python bench.py --no-cpu-baseline --batch 2 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
echo "=== bench B=2 serial"; timeout 600 python bench.py --no-cpu-baseline --batch 2 --steps 30 --serial-backward 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
