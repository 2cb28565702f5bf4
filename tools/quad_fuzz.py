"""Randomised cross-check of the device quad-tree build (not part of the test suite: run on a GPU box): random sizes,
distributions and precisions; the device-built tree against the host builder's, array by array (the host builder inserts point
after point, as quad_tree.rs does), and three steps against three steps with the host builder (NBODY_TREE_BUILD_HOST=1).
    python tools/quad_fuzz.py [cases=120] [seed=1] [log10 of the smallest size=0] [of the largest=5.6]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
lo_exp = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi_exp = float(sys.argv[4]) if len(sys.argv) > 4 else 5.6


def scene(kind, n, dt):
    if kind == 0: return (rng.random((n, 2)) * 9e4 + 5e3).astype(dt)
    if kind == 1: return (rng.standard_normal((n, 2)) * 1.2e4 + 5e4).astype(dt)          # a few outside the root cell
    if kind == 2: return (10.0 ** rng.uniform(0, 4.9, (n, 2))).astype(dt)               # crowded towards a corner
    if kind == 3: return (rng.integers(0, 2000, (n, 2)) * 37.5 + 100).astype(dt)         # a lattice: coincident points, full leaves
    return nb.scenes.plummer(n, seed=int(rng.integers(1, 1 << 30)), dtype=dt)[0]


bad = 0
with C.Context(0) as ctx:
    for case in range(cases):
        n = int(10 ** rng.uniform(lo_exp, hi_exp))
        dt = np.float64 if rng.integers(0, 2) else np.float32
        kind = int(rng.integers(0, 5))
        pos = scene(kind, n, dt)
        w = rng.integers(1, 9, n).astype(np.uint32)
        tag = f"case {case}: n {n} {np.dtype(dt).name} kind {kind}"
        h = C.host_tree(C.TREE_QUAD, pos, w, C.default_params())
        if h["overflow"]:
            print(tag, "degenerate for the reference itself: skipped"); continue
        ctx.set_params(theta=0.5)
        ctx.upload(pos, np.zeros_like(pos), w)
        ctx.accel_tree(C.TREE_QUAD, pos[:1])
        t = ctx.tree_export()
        ok = all(np.array_equal(t[k], h[k]) for k in ("mass", "is_leaf", "first", "count", "skip", "order")) and \
            np.array_equal(t["geom"], h["geom"], equal_nan=True)
        res = []
        vel = (rng.standard_normal((n, 2)) * 10).astype(dt)
        same = True
        try:
            for host in ("0", "1"):
                os.environ["NBODY_TREE_BUILD_HOST"] = host
                ctx.upload(pos, vel, w)
                ctx.update_tree(C.TREE_QUAD, 0.05, 3)
                res.append(ctx.download())
            same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*res))
        except C.NBodyError as e:
            print(tag, "steps:", str(e)[:60])
        os.environ.pop("NBODY_TREE_BUILD_HOST", None)
        if not (ok and same): bad += 1
        print(tag, "tree", "ok" if ok else "MISMATCH", "steps", "ok" if same else "MISMATCH", flush=True)
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
