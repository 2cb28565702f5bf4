cd $GRAFT_REPO_ROOT
echo "== packed two-target exact rounds"; python tools/walk_fast_ab.py bvh
python tools/ref_scene_steps.py 300 20
python -m pytest tests/test_gpu_tree.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_gpu_differential.py -x -q 2>&1 | tail -3
