cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
HBIRD_FUZZ_ROWS=200000 timeout 900 python tests/fuzz_small.py 60 11 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -1
timeout 600 python bench.py --fp16 2>&1 | tail -1
