#!/bin/bash
# round 3: the rocprofv3 passes of the direct legs (run on the GPU box from the repo root); summaries land in gpurun_out/profiles_new/
set -e
mkdir -p gpurun_out/profiles_new
P=gpurun_out/profiles_new
bash tools/profile_legs.sh headline 'direct_stream' $P/r03_direct_pmc.json && echo headline done
bash tools/profile_legs.sh config2 'direct_stream' $P/r03_leg_config2_pmc.json && echo config2 done
bash tools/profile_legs.sh per_body_masses 'direct_stream' $P/r03_leg_per_body_masses_pmc.json && echo per_body done
bash tools/profile_legs.sh reference_scene_direct 'direct_stream' $P/r03_leg_reference_scene_direct_pmc.json && echo refdirect done
