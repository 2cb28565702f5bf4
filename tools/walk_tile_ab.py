"""Three-pass walk (NBODY_WALK_SPLIT=4) against the one-pass LDS walk (default) and the fused walk (=0): same bits, phase times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi


def run(pos, vel, w, env, steps=6, theta=50.0, leaf=64):
    for k in list(os.environ):
        if k.startswith("NBODY_WALK"):
            del os.environ[k]
    os.environ.update(env)
    with C.Context(0) as ctx:
        ctx.set_params(theta=theta, leaf_size=leaf)
        ctx.upload(pos, vel, w)
        cnt = C.Counting()
        ctx.update_tree(C.TREE_BVH, 0.1, 2, cnt)
        cnt = C.Counting()
        t0 = time.perf_counter()
        ctx.update_tree(C.TREE_BVH, 0.1, steps, cnt)
        dt = (time.perf_counter() - t0) / steps
        p, v, _, ids = ctx.download()
    return p, v, ids, 1e3 * dt, 1e3 * cnt.build_bvh / steps, 1e3 * cnt.sum_gravity / steps


cases = [("reference scene", nb.scenes.galaxy(), 50.0), ("plummer 1M", nb.scenes.plummer(1 << 20, seed=0x5EED0002, dtype=np.float32), 50.0),
         ("plummer 4M", nb.scenes.plummer(1 << 22, seed=0x5EED0003, dtype=np.float32), 50.0),
         ("plummer 256k", nb.scenes.plummer(1 << 18, seed=7, dtype=np.float32), 50.0),
         ("plummer 256k th 5", nb.scenes.plummer(1 << 18, seed=5, dtype=np.float32), 5.0),
         ("plummer 64k th 0.5", nb.scenes.plummer(1 << 16, seed=6, dtype=np.float32), 0.5),
         ("uniform 1M", ((np.random.default_rng(1).random((1 << 20, 2)) * 1e5).astype(np.float32), np.zeros((1 << 20, 2), np.float32), np.ones(1 << 20, np.uint32)), 50.0)]
variants = [("three passes (NBODY_WALK_SPLIT=4)", {"NBODY_WALK_SPLIT": "4"}), ("one pass (default)", {}), ("fused", {"NBODY_WALK_SPLIT": "0"})]
for name, (pos, vel, w), theta in cases:
    ref = None
    for vname, env in variants:
        p, v, ids, ms, b, wk = run(pos, vel, w, env, theta=theta)
        same = "" if ref is None else ("  bits equal" if (np.array_equal(p.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(v.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(ids, ref[2])) else "  BITS DIFFER")
        if ref is None:
            ref = (p, v, ids)
        print(f"{name:18s} {vname:36s} step {ms:8.3f} ms  build {b:6.3f}  walk {wk:8.3f}{same}", flush=True)
