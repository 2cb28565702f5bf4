# Round 4: the leaf-step arm thresholds of the one-pass walks (lane = particle rounds vs lane = target) swept on the bench's BVH legs:
# the laboratory library rebuilt on the GPU box with -DNB_TILE_ROUND_COST / -DNB_FAST_ROUND_COST.   bash tools/walk_round_cost_sweep.sh
cd nbody-simulation_amd/csrc
export NBODY_HIP_LIBRARY=lab
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for cfg in "" "-DNB_TILE_ROUND_COST=40 -DNB_FAST_ROUND_COST=12" "-DNB_TILE_ROUND_COST=52 -DNB_FAST_ROUND_COST=16" "-DNB_TILE_ROUND_COST=85 -DNB_FAST_ROUND_COST=24" "-DNB_TILE_ROUND_COST=110 -DNB_FAST_ROUND_COST=32"; do
  rm -f walk_split.lab.o
  make lab CXXFLAGS="$BASE $cfg" > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  for leg in reference_scene_bvh plummer1m_bvh; do
  (cd ../.. && timeout -k 10 150 python bench.py --leg $leg --no-cpu-baseline > gpurun_out/r04_rc.json 2> gpurun_out/r04_rc.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r04_rc.json'))
print(repr(sys.argv[1]), sys.argv[2], 'exact kernel', round(d['exact']['roofline']['kernel_ms'],4), 'FAST kernel', round(d['fast']['roofline']['kernel_ms'],4))
" "$cfg" $leg) || echo "run failed: $cfg"
  done
done
