#!/bin/bash
# round 4: the rocprofv3 passes of the tree legs (run on the GPU box from the repo root); see profile_all_direct.sh
set -e
P=profiles
bash tools/profile_legs.sh reference_scene_bvh 'walk_tile<' $P/r04_leg_reference_scene_bvh_pmc.json 'walk_tile_fast' $P/r04_leg_reference_scene_bvh_fast_pmc.json && echo refbvh done
bash tools/profile_legs.sh plummer1m_bvh 'walk_tile<' $P/r04_leg_plummer1m_bvh_pmc.json 'walk_tile_fast' $P/r04_leg_plummer1m_bvh_fast_pmc.json && echo plummer done
bash tools/profile_legs.sh config4 'tree_walk_small' $P/r04_leg_config4_pmc.json 'tree_walk_wave' $P/r04_leg_config4_fast_pmc.json && echo config4 done
mkdir -p gpurun_out/profiles_new && cp $P/r04_* gpurun_out/profiles_new/
