"""A/B of the two tree-walk kernels (per-thread vs wave-uniform), same process, same trees."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi

VARIANTS = [("per-thread exact", {"NBODY_WALK_PER_THREAD": "1", "ARITH": "0"}), ("wave exact", {"NBODY_WALK_PER_THREAD": "0", "ARITH": "0"}),
            ("wave FAST", {"NBODY_WALK_PER_THREAD": "0", "ARITH": "1"})]


def run(name, pos, vel, w, kind, theta, order=C.ORDER_CONSISTENT):
    res = {}
    for vname, env in VARIANTS * 2:
        os.environ["NBODY_WALK_PER_THREAD"] = env["NBODY_WALK_PER_THREAD"]
        with C.Context(0) as ctx:
            ctx.set_params(theta=theta, order=order, arith=int(env["ARITH"]))
            ctx.upload(pos, vel, w)
            ctx.update_tree(kind, 0.1, 1)
            t = C.Timer(); ctx.set_timer(t)
            ctx.update_tree(kind, 0.1, 3)
            ms, _ = t.read()
            res.setdefault(vname, []).append(ms)
    print(name + ": " + " | ".join(f"{k} {min(v):.3f}" for k, v in res.items()), flush=True)

pos, vel, w = nb.scenes.galaxy()
run("reference scene bvh theta 50 (as written)", pos, vel, w, C.TREE_BVH, 50.0, C.ORDER_AS_WRITTEN)
run("reference scene bvh theta 50 (consistent)", pos, vel, w, C.TREE_BVH, 50.0)
run("reference scene quad theta 0.5", pos, vel, w, C.TREE_QUAD, 0.5)
pos, vel, w = nb.scenes.plummer(1 << 20, seed=0x5EED0003)
run("plummer 1M quad theta 0.5 f32", pos, vel, w, C.TREE_QUAD, 0.5)
run("plummer 1M bvh theta 50 f32", pos, vel, w, C.TREE_BVH, 50.0)
pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
run("config 4: plummer 4M quad theta 0.5 f64", pos, vel, w, C.TREE_QUAD, 0.5)
