"""NB_WALK_TIMING build only: time of the longest wave of the reference scene's BVH walk, split leaf / node steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
pos, vel, w = nb.scenes.galaxy()
for arith in (0, 1):
    with C.Context(0) as ctx:
        ctx.set_params(theta=50.0, order=C.ORDER_CONSISTENT, arith=arith)
        ctx.upload(pos, vel, w)
        ctx.update_tree(C.TREE_BVH, 0.1, 1)
        ctx.walk_stats(True)
        ctx.accel_tree(C.TREE_BVH)
        a, b, tot = ctx.walk_stats(False)
        print(f"arith {arith}: longest wave {(a >> 40) * 0.01:.1f} us = leaf steps {((a >> 20) & 0xFFFFF) * 0.01:.1f} us + node steps {(a & 0xFFFFF) * 0.01:.1f} us;"
              f" it made {(b >> 20) & 0xFFFFF} leaf steps, {b & 0xFFFFF} node steps; all waves together {tot * 0.01 / 1e3:.1f} ms")
