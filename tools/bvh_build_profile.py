"""A few BVH steps of one scene, for rocprofv3 --kernel-trace --stats (per-kernel time of the device build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
which = sys.argv[1] if len(sys.argv) > 1 else "galaxy"
pos, vel, w = nb.scenes.galaxy() if which == "galaxy" else nb.scenes.plummer(int(which), seed=0x5EED0003)
with C.Context(0) as ctx:
    ctx.set_params(theta=50.0)
    ctx.upload(pos, vel, w)
    ctx.update_tree(C.TREE_BVH, 0.1, 10)
    print("on device:", ctx.last_build_on_device(), "restarts", ctx.bvh_build_restarts())
