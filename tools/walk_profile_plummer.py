"""Plummer N = 1 048 576, BVH, theta 50: a few steps, for rocprofv3 (NBODY_TRACE=1 prints the term counts)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0002, dtype=np.float32)
with C.Context(0) as ctx:
    ctx.set_params(theta=50.0)
    ctx.upload(pos, vel, w)
    cnt = C.Counting()
    ctx.update_tree(C.TREE_BVH, 0.1, 1, cnt)
    cnt = C.Counting()
    ctx.update_tree(C.TREE_BVH, 0.1, 4, cnt)
    print("build %.2f ms walk %.2f ms integrate %.3f ms per step" % (cnt.build_bvh / 4 * 1e3, cnt.sum_gravity / 4 * 1e3, cnt.post_calculations / 4 * 1e3))
