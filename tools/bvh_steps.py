"""BVH (theta 50, the reference's setting) steps on the reference scene and on Plummer spheres: ms/step and the Counting split.
    python tools/bvh_steps.py [quick]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi


def run(name, pos, vel, w, steps, warm, kind=C.TREE_BVH, theta=50.0):
    with C.Context(0) as c:
        c.set_params(theta=theta)
        c.upload(pos, vel, w)
        c.update_tree(kind, 0.1, warm)
        cnt = C.Counting()
        t0 = time.perf_counter()
        c.update_tree(kind, 0.1, steps, cnt)
        dt = time.perf_counter() - t0
    print(json.dumps({"case": name, "n": int(pos.shape[0]), "steps": steps, "ms_per_step": round(1e3 * dt / steps, 4),
                      "build_ms": round(1e3 * cnt.build_bvh / steps, 4), "walk_ms": round(1e3 * cnt.sum_gravity / steps, 4),
                      "integrate_ms": round(1e3 * cnt.post_calculations / steps, 4)}), flush=True)


pos, vel, w = nb.scenes.galaxy()
run("reference scene, BVH theta 50", pos, vel, w, 30, 5)
pos, vel, w = nb.scenes.plummer(1 << 20, seed=0x5EED0003)
run("Plummer 1M, BVH theta 50", pos, vel, w, 10, 3)
if len(sys.argv) < 2:
    run("Plummer 1M, quad theta 0.5", pos, vel, w, 10, 3, C.TREE_QUAD, 0.5)
    pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
    run("config 4: Plummer 4M, quad theta 0.5, f64", pos, vel, w, 5, 2, C.TREE_QUAD, 0.5)
    run("Plummer 4M, BVH theta 50, f64", pos, vel, w, 3, 1)
