"""A/B of the Barnes-Hut walks: the exact arm (bit parity) against the tolerance-contract FAST arm, whole steps.
One JSON line per case and arithmetic: ms/step, Counting split, the walk kernel's time (HIP events)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402

C = nb._capi


def case(name, pos, vel, w, kind, theta, steps):
    for label, arith in (("exact", C.ARITH_AUTO), ("fast", C.ARITH_FAST)):
        with C.Context(0) as ctx:
            ctx.set_params(theta=theta, order=C.ORDER_AS_WRITTEN if kind == C.TREE_BVH else C.ORDER_CONSISTENT, arith=arith)
            ctx.upload(pos, vel, w)
            ctx.update_tree(kind, 0.1, 2)
            t = C.Timer()
            ctx.set_timer(t)
            cnt = C.Counting()
            t0 = time.perf_counter()
            ctx.update_tree(kind, 0.1, steps, cnt)
            wall = time.perf_counter() - t0
            kms, kl = t.read()
            ctx.set_timer(None)
        print(json.dumps({"case": name, "arith": label, "n": pos.shape[0], "steps": steps, "ms_per_step": 1e3 * wall / steps,
                          "build_ms": 1e3 * cnt.build_bvh / steps, "walk_phase_ms": 1e3 * cnt.sum_gravity / steps,
                          "integrate_ms": 1e3 * cnt.post_calculations / steps, "walk_kernel_ms": kms, "launches": kl}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    pos, vel, w = nb.scenes.galaxy()
    case("reference scene, BVH theta 50", pos, vel, w, C.TREE_BVH, 50.0, 100)
    pos, vel, w = nb.scenes.plummer(1 << 20, seed=0x5EED0003)
    case("Plummer 1M, BVH theta 50", pos, vel, w, C.TREE_BVH, 50.0, 10)
    if which == "all":
        case("Plummer 1M, quad theta 0.5", pos, vel, w, C.TREE_QUAD, 0.5, 10)
        pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
        case("config 4: Plummer 4M, quad theta 0.5, f64", pos, vel, w, C.TREE_QUAD, 0.5, 5)
        case("Plummer 4M, BVH theta 50, f64", pos, vel, w, C.TREE_BVH, 50.0, 3)
