"""Diagnostic counters of the device BVH build, one plain (not enqueued-ahead) step per scene: scan restarts, prepared runs used
and, in a library built with -DNB_FOLD_TIMING / -DNB_BVH_TIMING (make CXXFLAGS="... -DNB_FOLD_TIMING"), the per-level split of
bvh_big_fold (real adds / scan rounds / slow rounds) and the slowest group per phase of bvh_subtrees.
    python tools/fold_timing.py"""
import os, sys
os.environ["NBODY_TRACE"] = "1"
os.environ["NBODY_STEP_AHEAD"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
for name, (pos, vel, w) in (("reference scene", nb.scenes.galaxy()), ("Plummer 1M", nb.scenes.plummer(1 << 20, seed=0x5EED0003))):
    print(name, "x in", pos[:, 0].min(), pos[:, 0].max(), file=sys.stderr)
    with C.Context(0) as c:
        c.set_params(theta=50.0)
        c.upload(pos, vel, w)
        c.update_tree(C.TREE_BVH, 0.1, 1)
