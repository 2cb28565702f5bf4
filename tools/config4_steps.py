import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
with C.Context(0) as ctx:
    ctx.set_params(theta=0.5)
    ctx.upload(pos, vel, w)
    ctx.update_tree(C.TREE_QUAD, 0.1, 1)
    cnt = C.Counting()
    ctx.update_tree(C.TREE_QUAD, 0.1, 5, cnt)
    print("build %.3f ms walk %.3f ms integrate %.3f ms per step" % (cnt.build_bvh / 5 * 1e3, cnt.sum_gravity / 5 * 1e3, cnt.post_calculations / 5 * 1e3))
