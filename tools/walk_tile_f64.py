"""f64 BVH walk: fused (NBODY_WALK_SPLIT=0) against the one-pass LDS walk with 4 or 8 tile rows: same bits, phase times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi


def run(pos, vel, w, env, steps=3, theta=50.0):
    for k in list(os.environ):
        if k.startswith("NBODY_WALK"):
            del os.environ[k]
    os.environ.update(env)
    with C.Context(0) as ctx:
        ctx.set_params(theta=theta)
        ctx.upload(pos, vel, w)
        ctx.update_tree(C.TREE_BVH, 0.1, 2)
        cnt = C.Counting()
        t0 = time.perf_counter()
        ctx.update_tree(C.TREE_BVH, 0.1, steps, cnt)
        dt = (time.perf_counter() - t0) / steps
        p, v, _, ids = ctx.download()
    return p, v, ids, 1e3 * dt, 1e3 * cnt.build_bvh / steps, 1e3 * cnt.sum_gravity / steps


g = nb.scenes.galaxy(dtype=np.float64)
cases = [("reference scene f64", g, 50.0), ("plummer 1M f64", nb.scenes.plummer(1 << 20, seed=0x5EED0002, dtype=np.float64), 50.0),
         ("plummer 4M f64", nb.scenes.plummer(1 << 22, seed=0x5EED0003, dtype=np.float64), 50.0)]
variants = [("fused", {"NBODY_WALK_SPLIT": "0"}), ("one pass, 8 rows", {"NBODY_WALK_TILE_TARGETS": "8"}), ("one pass, 4 rows", {"NBODY_WALK_TILE_TARGETS": "4"})]
for name, (pos, vel, w), theta in cases:
    ref = None
    for vname, env in variants:
        p, v, ids, ms, b, wk = run(pos, vel, w, env, theta=theta)
        same = "" if ref is None else ("  bits equal" if (np.array_equal(p.view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(v.view(np.uint64), ref[1].view(np.uint64)) and np.array_equal(ids, ref[2])) else "  BITS DIFFER")
        if ref is None:
            ref = (p, v, ids)
        print(f"{name:22s} {vname:20s} step {ms:9.3f} ms  build {b:7.3f}  walk {wk:9.3f}{same}", flush=True)
