"""A few BVH steps (theta 50) of a Plummer sphere, for a kernel trace: python tools/plummer_bvh_steps.py [steps=8] [n=1<<20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
with C.Context(0) as c:
    c.set_params(theta=50.0)
    c.upload(pos, vel, w)
    c.update_tree(C.TREE_BVH, 0.1, 3)
    cnt = C.Counting()
    c.update_tree(C.TREE_BVH, 0.1, steps, cnt)
    print("build %.3f ms walk %.3f ms per step" % (cnt.build_bvh / steps * 1e3, cnt.sum_gravity / steps * 1e3))
