"""Reference scene (World::new, seeded), BVH, theta 50: a few steps, for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
pos, vel, w = nb.scenes.galaxy()
with C.Context(0) as ctx:
    ctx.set_params(theta=50.0)
    ctx.upload(pos, vel, w)
    cnt = C.Counting()
    ctx.update_tree(C.TREE_BVH, 0.1, 6, cnt)
    print("build %.2f ms walk %.2f ms integrate %.3f ms per step" % (cnt.build_bvh / 6 * 1e3, cnt.sum_gravity / 6 * 1e3, cnt.post_calculations / 6 * 1e3))
