"""Plummer 4M f64 BVH: a few builds (accel_tree on 4 targets), for rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0004, dtype=np.float64)
with C.Context(0) as c:
    c.set_params(theta=50.0)
    c.upload(pos, vel, w)
    for _ in range(4):
        c.accel_tree(C.TREE_BVH, pos[:4])
    print("device", c.last_build_on_device(), "nodes", c.tree_info().n_nodes, "depth", c.tree_info().max_depth)
