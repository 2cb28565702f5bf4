"""A few quad-tree steps (Plummer 1 M, f32, theta 0.5), for a kernel trace: python tools/quad_steps.py [steps=10] [n=1<<20] [f64]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003, dtype=dt)
with C.Context(0) as ctx:
    ctx.set_params(theta=0.5)
    ctx.upload(pos, vel, w)
    ctx.update_tree(C.TREE_QUAD, 0.1, 2)
    cnt = C.Counting()
    ctx.update_tree(C.TREE_QUAD, 0.1, steps, cnt)
    print("build %.3f ms walk %.3f ms integrate %.3f ms per step" % (cnt.build_bvh / steps * 1e3, cnt.sum_gravity / steps * 1e3, cnt.post_calculations / steps * 1e3))
