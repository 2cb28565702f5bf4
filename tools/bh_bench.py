"""Barnes-Hut step timings (secondary path; the headline bench is bench.py).  Prints one JSON line per case:
phase seconds as the reference's Counting names them, walk statistics, the walk priced in flops against the VALU peak
(the walks are VALU-bound on the reference's two IEEE divisions per pair — profiles/r02_leg_config4_pmc.json,
r01_walk_tile_pmc.json; round 1 priced bytes against HBM, which over-counts by the wave-sharing factor: node and leaf
data reach a wave by scalar loads shared by its 64 targets) and the CPU oracle's timing of the same step on the host cores."""
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


def case(name, pos, vel, w, kind, theta, steps=3, cpu=True, cpu_steps=1):
    from oracle import oracle as orc
    f64 = pos.dtype == np.float64
    node_bytes = (32 + 32 + 16) if not f64 else (64 + 64 + 16)
    pair_bytes = 12 if not f64 else 24
    with C.Context(0) as ctx:
        ctx.set_params(theta=theta, order=C.ORDER_CONSISTENT)
        ctx.upload(pos, vel, w)
        ctx.update_tree(kind, 0.1, 1)                     # warm-up (allocations, first build)
        t = C.Timer()
        ctx.set_timer(t)
        cnt = C.Counting()
        t0 = time.perf_counter()
        ctx.update_tree(kind, 0.1, steps, cnt)
        wall = time.perf_counter() - t0
        walk_ms, launches = t.read()
        ctx.set_timer(None)
        ctx.walk_stats(True)                              # statistics in a separate, untimed walk (3 atomics per target)
        ctx.accel_tree(kind)
        visits, accepted, leaf_pairs = ctx.walk_stats(False)
        info = ctx.tree_info()
    n = pos.shape[0]
    # lower bound of the walk's bytes: per-target node and leaf bytes / 64 (wave-uniform scalar loads) + targets in, accelerations out
    alg_bytes = (visits * node_bytes + leaf_pairs * pair_bytes) / 64 + n * (2 * (8 if f64 else 4)) * 2
    flops = 14.0 * (accepted + leaf_pairs) + 12.0 * visits        # as bench.py: pair evaluation 14 (a division = 1), node test 12
    peak = 78.6 if f64 else 157.3
    out = {"case": name, "n": n, "dtype": "f64" if f64 else "f32", "tree": "bvh" if kind == C.TREE_BVH else "quad",
           "theta": theta, "nodes": info.n_nodes, "max_depth": info.max_depth, "steps": steps,
           "ms_per_step": 1e3 * wall / steps, "build_ms": 1e3 * cnt.build_bvh / steps,
           "walk_phase_ms": 1e3 * cnt.sum_gravity / steps, "integrate_ms": 1e3 * cnt.post_calculations / steps,
           "walk_kernel_ms": walk_ms, "node_visits_per_target": visits / n, "leaf_pairs_per_target": leaf_pairs / n,
           "accepted_per_target": accepted / n, "walk_bound": "valu_f64" if f64 else "valu_f32",
           "walk_TFLOPs": flops / (walk_ms * 1e-3) / 1e12 if walk_ms > 0 else None,
           "walk_frac_of_vector_peak": flops / (walk_ms * 1e-3) / 1e12 / peak if walk_ms > 0 else None,
           "walk_bytes_lower_bound_GB": alg_bytes / 1e9,
           "walk_bytes_lower_bound_GBps": alg_bytes / 1e9 / (walk_ms * 1e-3) if walk_ms > 0 else None,
           "interactions_per_s": (accepted + leaf_pairs) / (walk_ms * 1e-3) if walk_ms > 0 else None}
    if cpu:
        cores = min(len(os.sched_getaffinity(0)), 64)
        try:
            orc.build(native=True)
            native = True
        except Exception:
            native = False
        t0 = time.perf_counter()
        if kind == C.TREE_BVH:
            _, _, _, _, c3 = orc.update_bvh(pos, vel, w, delta=0.1, theta=theta, mode=orc.CONSISTENT, nsteps=cpu_steps,
                                            nthreads=cores, native_lib=native)
        else:
            _, _, c3 = orc.update_quad(pos, vel, w, delta=0.1, theta=theta, nsteps=cpu_steps, nthreads=cores, native_lib=native)
        dt = time.perf_counter() - t0
        out["cpu_oracle"] = {"threads": cores, "ms_per_step": 1e3 * dt / cpu_steps, "build_ms": 1e3 * c3[0] / cpu_steps,
                             "walk_ms": 1e3 * c3[1] / cpu_steps, "integrate_ms": 1e3 * c3[2] / cpu_steps}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    pos, vel, w = nb.scenes.galaxy()
    case("reference scene (World::new, seeded), theta 50", pos, vel, w, C.TREE_BVH, 50.0, steps=5)
    if quick:
        sys.exit(0)
    case("reference scene, theta 0.5", pos, vel, w, C.TREE_BVH, 0.5, steps=5)
    pos, vel, w = nb.scenes.plummer(1 << 20, seed=0x5EED0003)
    case("plummer 1M bvh theta 50 (the reference's theta)", pos, vel, w, C.TREE_BVH, 50.0, steps=3)
    case("plummer 1M bvh theta 0.5 (needle boxes: ~direct sum)", pos, vel, w, C.TREE_BVH, 0.5, steps=1, cpu=False)
    case("plummer 1M quad theta 0.5", pos, vel, w, C.TREE_QUAD, 0.5, steps=3)
    pos, vel, w = nb.scenes.plummer(1 << 22, seed=0x5EED0004, dtype=np.float64)
    case("config 4: plummer 4M quad theta 0.5 f64", pos, vel, w, C.TREE_QUAD, 0.5, steps=3)
    case("plummer 4M bvh theta 50 f64", pos, vel, w, C.TREE_BVH, 50.0, steps=3)
