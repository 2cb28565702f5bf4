import os, sys
os.environ["NBODY_TRACE"] = "1"; os.environ["NBODY_STEP_AHEAD"] = "0"
sys.path.insert(0, "/root/repo")
import numpy as np
import nbody_simulation_amd as nb
C = nb._capi
for name, (pos, vel, w) in (("galaxy", nb.scenes.galaxy()), ("plummer1M", nb.scenes.plummer(1 << 20, seed=0x5EED0003))):
    print(name, "x range", pos[:, 0].min(), pos[:, 0].max(), "mean", pos[:, 0].mean(), file=sys.stderr)
    with C.Context(0) as c:
        c.set_params(theta=50.0)
        c.upload(pos, vel, w)
        c.update_tree(C.TREE_BVH, 0.1, 1)
        r = c.bvh_build_restarts() if hasattr(c, "bvh_build_restarts") else -1
        print(name, "stops", r & 0xffff, "rounds", r >> 16, file=sys.stderr)
