cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; timeout 600 python -m pytest tests/test_tail_split_gpu.py -m gpu -q --durations=5 2>&1 | tail -12
