"""Is the reference scene's BVH walk bound by its longest wave?  Times the FAST walk (count + scan + walk kernels) for all
targets, for the 4 096 targets closest to the heaviest body (the long walks), and for 4 096 random ones."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
os.environ["NBODY_WALK_SPLIT"] = "3"
pos, vel, w = nb.scenes.galaxy()
d = np.abs(pos - pos[np.argmax(w)]).sum(axis=1)
near = pos[np.argsort(d)[:4096]]
rnd = pos[np.random.default_rng(1).choice(pos.shape[0], 4096, replace=False)]
for arith, name in ((C.ARITH_AUTO, "exact"), (C.ARITH_FAST, "fast")):
    with C.Context(0) as ctx:
        ctx.set_params(theta=50.0, order=C.ORDER_CONSISTENT, arith=arith)
        ctx.upload(pos, vel, w)
        ctx.accel_tree(C.TREE_BVH, pos[:4096])
        for label, tg in (("all", pos), ("4096 nearest the heavy body", near), ("4096 random", rnd), ("64 nearest", near[:64]), ("1 nearest", near[:1])):
            t = C.Timer()
            ctx.set_timer(t)
            for _ in range(5):
                ctx.accel_tree(C.TREE_BVH, tg)
            ms, k = t.read()
            ctx.set_timer(None)
            print(f"{name}: {label}: {ms:.3f} ms per walk (count + scan + walk kernels, {k} timed)")
