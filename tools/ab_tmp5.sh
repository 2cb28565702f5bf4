cd $GRAFT_REPO_ROOT
python tools/walk_fast_ab.py all
python -m pytest tests/test_gpu_tree.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_differential.py tests/test_gpu_walk_fast.py -x -q 2>&1 | tail -3
