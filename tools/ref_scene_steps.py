"""The reference's own workload — World::new's scene, BVH, theta 50, dt 0.1 (main.rs:31-35) — stepped n times in one
call; prints ms/step and the Counting split.    python tools/ref_scene_steps.py [steps=200] [warmup=20]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pos, vel, w = nb.scenes.galaxy()
with C.Context(0) as c:
    c.upload(pos, vel, w)
    c.update_tree(C.TREE_BVH, 0.1, warm)
    cnt = C.Counting()
    t0 = time.perf_counter()
    c.update_tree(C.TREE_BVH, 0.1, steps, cnt)
    dt = time.perf_counter() - t0
print(json.dumps({"scene": "World::new (seeded)", "n": int(pos.shape[0]), "steps": steps, "ms_per_step": 1e3 * dt / steps,
                  "build_ms": 1e3 * cnt.build_bvh / steps, "walk_ms": 1e3 * cnt.sum_gravity / steps,
                  "integrate_ms": 1e3 * cnt.post_calculations / steps, "step_ahead": os.environ.get("NBODY_STEP_AHEAD", "1")}))
