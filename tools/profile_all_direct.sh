#!/bin/bash
# round 4: the rocprofv3 passes of the direct legs (run on the GPU box from the repo root); summaries land in gpurun_out/profiles_new/
# AND in profiles/ of the box's copy (bench.py reads the traffic figure from there); copy gpurun_out/profiles_new/* to profiles/ afterwards
set -e
P=profiles
bash tools/profile_legs.sh headline 'direct_stream' $P/r04_direct_pmc.json && echo headline done
bash tools/profile_legs.sh config2 'direct_stream' $P/r04_leg_config2_pmc.json && echo config2 done
bash tools/profile_legs.sh mass_classes 'direct_stream' $P/r04_leg_mass_classes_pmc.json && echo mass_classes done
bash tools/profile_legs.sh free_masses 'direct_stream_m' $P/r04_leg_free_masses_pmc.json && echo free_masses done
bash tools/profile_legs.sh reference_scene_direct 'direct_stream' $P/r04_leg_reference_scene_direct_pmc.json && echo refdirect done
mkdir -p gpurun_out/profiles_new && cp $P/r04_* gpurun_out/profiles_new/
