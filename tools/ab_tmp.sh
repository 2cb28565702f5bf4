set -e
cd $GRAFT_REPO_ROOT
for m in 2 3 2 3; do echo "ASM=$m"; NBODY_DIRECT_ASM=$m python bench.py --no-legs --no-cpu-baseline --steps 5 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -5
