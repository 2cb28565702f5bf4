set -o pipefail
cd nbody-simulation_amd/csrc
export NBODY_HIP_LIBRARY=lab
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden"
for cfg in "" "-DNB_CHAIN_START=256" "-DNB_CHAIN_START=1024" "-DNB_CHAIN_START=2048" "-DNB_RUN_LEN=8192" "-DNB_RUN_LEN=32768" "-DNB_RUN_LEN=1000000" "-DNB_KSUB=1024" "-DNB_KSUB=4096"; do
  rm -f bvh_build.lab.o
  make lab CXXFLAGS="$BASE $cfg" > /dev/null 2>&1 || { echo "build failed: $cfg"; continue; }
  (cd ../.. && timeout -k 10 120 python bench.py --leg reference_scene_bvh --no-cpu-baseline > gpurun_out/r04_bs.json 2> gpurun_out/r04_bs.err && python -c "
import json,sys; d=json.load(open('gpurun_out/r04_bs.json'))
print(repr(sys.argv[1]), 'build_ms', round(d['exact']['build_ms'],4), 'step', round(d['exact']['ms_per_step'],4), 'fast step', round(d['fast']['ms_per_step'],4))
" "$cfg") || echo "run failed: $cfg"
done
