set -e
cd $GRAFT_REPO_ROOT
for cs in 256 512 1024 2048; do
  cd nbody-simulation_amd/csrc && touch bvh_build.hip && make -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DNB_CHAIN_START=$cs" > /dev/null 2>&1 && cd ../..
  echo "chain start $cs"; python tools/bvh_steps.py quick 2>/dev/null
done
