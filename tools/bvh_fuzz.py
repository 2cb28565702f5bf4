"""Randomised cross-check of the device BVH build and the step enqueued ahead of the host (not part of the test suite: run on a
GPU box).  Every case: a random size, leaf size and distribution; the device-built tree against the host builder's, array by
array, and four steps enqueued ahead against four phase-by-phase steps, row by row.
    python tools/bvh_fuzz.py [cases=150] [seed=1] [log10 of the smallest size=0] [of the largest=5.6] [f64]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo_exp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi_exp = float(sys.argv[4]) if len(sys.argv) > 4 else 5.6
F32 = np.float64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else np.float32


def scene(kind, n):
    if kind == 0: return (rng.random((n, 2)) * 1e5).astype(F32)
    if kind == 1: return (rng.standard_normal((n, 2)) * 3e4).astype(F32)
    if kind == 2: return (-rng.random((n, 2)) * 1e5).astype(F32)
    if kind == 3: return (10.0 ** rng.uniform(-6, 6, (n, 2))).astype(F32)
    if kind == 4: return (rng.integers(0, 3000, (n, 2)) * 0.5).astype(F32)
    return nb.scenes.plummer(n, seed=int(rng.integers(1, 1 << 30)), dtype=F32)[0]


bad = 0
with C.Context(0) as ctx:
    for case in range(cases):
        n = int(10 ** rng.uniform(lo_exp, hi_exp))
        leaf = int(rng.choice([1, 4, 16, 64, 64, 64, 200, 1000]))
        kind = int(rng.integers(0, 6))
        pos = scene(kind, n)
        if leaf < 16 and n > 60000: n = 60000; pos = pos[:n].copy()
        w = rng.integers(1, 9, n).astype(np.uint32)
        prm = C.default_params(); prm.leaf_size = leaf
        h = C.host_tree(C.TREE_BVH, pos, w, prm)
        tag = f"case {case}: n {n} leaf {leaf} kind {kind}"
        if h["overflow"]:
            print(tag, "degenerate for the reference itself: skipped"); continue
        ctx.set_params(theta=50.0, leaf_size=leaf)
        ctx.upload(pos, np.zeros_like(pos), w)
        ctx.accel_tree(C.TREE_BVH, pos[:1])
        dev = ctx.last_build_on_device()
        t = ctx.tree_export()
        ok = all(np.array_equal(t[k], h[k]) for k in ("mass", "is_leaf", "first", "count", "skip", "order")) and \
            np.array_equal(t["geom"], h["geom"], equal_nan=True)
        res = []
        vel = (rng.standard_normal((n, 2)) * 10).astype(F32)
        try:
            for ahead in ("1", "0"):
                os.environ["NBODY_STEP_AHEAD"] = ahead
                ctx.upload(pos, vel, w)
                ctx.update_tree(C.TREE_BVH, 0.05, 4)
                res.append(ctx.download())
            same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*res))
        except C.NBodyError as e:  # (points that come to coincide: the reference recurses without end, the library says so)
            same = len(res) == 0 or "depth cap" in str(e)
            print(tag, "steps:", str(e)[:60])
        os.environ.pop("NBODY_STEP_AHEAD", None)
        if not (ok and same): bad += 1
        print(tag, "device" if dev else "HOST", "tree", "ok" if ok else "MISMATCH", "steps", "ok" if same else "MISMATCH", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
