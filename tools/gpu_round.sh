set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-200
