"""Device vs host BVH build: `Counting` phases of full World::update steps, same process, same scenes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi


def run(name, pos, vel, w, steps=10):
    for host in ("0", "1"):
        os.environ["NBODY_TREE_BUILD_HOST"] = host
        with C.Context(0) as ctx:
            ctx.set_params(theta=50.0)
            ctx.upload(pos, vel, w)
            ctx.update_tree(C.TREE_BVH, 0.1, 2)
            cnt = nb.Counting()
            ctx.update_tree(C.TREE_BVH, 0.1, steps, cnt)
            info = ctx.tree_info()
            print(f"{name:28s} {'host  ' if host == '1' else 'device'} build {1e3 * cnt.build_bvh / steps:8.3f} ms  "
                  f"walk {1e3 * cnt.sum_gravity / steps:8.3f} ms  integrate {1e3 * cnt.post_calculations / steps:7.3f} ms  "
                  f"nodes {info.n_nodes} depth {info.max_depth} on_device {ctx.last_build_on_device()} "
                  f"restarts {ctx.bvh_build_restarts()}", flush=True)


pos, vel, w = nb.scenes.galaxy()
run("reference scene 151k", pos, vel, w)
for n in (1 << 14, 1 << 17, 1 << 20, 1 << 22):
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
    run(f"plummer {n}", pos, vel, w, steps=5 if n >= (1 << 20) else 10)
