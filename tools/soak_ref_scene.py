"""Soak of the reference's own workload against the oracle: World::new's scene (151 405 bodies), BVH, theta 50, dt 0.1, AS WRITTEN
(main.rs:388-425), stepped on the device (product library, steps enqueued ahead of the host) and by the CPU restatement, compared
bit for bit — positions, velocities, weights and the row permutation — every `every` steps.
    python tools/soak_ref_scene.py [steps=2000] [every=250] [bvh|quad] [f32|f64] [written|consistent]
A long run exercises what single steps do not: the walk's estimate from the previous walk's history, the device build's level
count learnt from the step before, the decoupled look-back scan's epochs, the speculation of the step enqueued ahead."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nbody_simulation_amd as nb  # noqa: E402  (the PRODUCT library: this is a check of what ships)
from oracle import oracle as orc  # noqa: E402
C = nb._capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
tree = sys.argv[3] if len(sys.argv) > 3 else "bvh"   # quad: the quad tree in World::update's place (quad_tree.rs:153-270; theta 0.5)
dtype = np.float64 if len(sys.argv) > 4 and sys.argv[4] == "f64" else np.float32
consistent = len(sys.argv) > 5 and sys.argv[5] == "consistent"   # every acceleration applied to the row it was computed for
pos, vel, w = nb.scenes.galaxy()
pos, vel = pos.astype(dtype), vel.astype(dtype)
n = pos.shape[0]
ids = np.arange(n, dtype=np.uint32)
o_pos, o_vel, o_w, o_ids = pos, vel, w, ids
out = {"scene": "World::new (seeded)", "n": int(n), "steps": steps, "compared_every": every, "tree": tree, "dtype": "f64" if dtype == np.float64 else "f32",
       "order": "consistent" if consistent else "as written", "checks": []}
with C.Context(0) as c:
    if tree == "quad":
        c.set_params(theta=0.5)
    if consistent:
        c.set_params(order=C.ORDER_CONSISTENT)
    c.upload(pos, vel, w)
    done = 0
    gpu_s = cpu_s = 0.0
    while done < steps:
        k = min(every, steps - done)
        t0 = time.perf_counter()
        gpu_degenerate = cpu_degenerate = False
        try:
            c.update_tree(C.TREE_QUAD if tree == "quad" else C.TREE_BVH, 0.1, k)
        except C.NBodyError as e:   # more points than a leaf holds at one place (or, quad: outside the root cell on one side): upstream recurses for ever
            if e.code != C.ERR_DEGENERATE:
                raise
            gpu_degenerate = True
        gpu_s += time.perf_counter() - t0
        t0 = time.perf_counter()
        try:
            if tree == "quad":
                o_pos, o_vel, _ = orc.update_quad(o_pos, o_vel, o_w, theta=0.5, nsteps=k, nthreads=16)   # (the quad build leaves the rows in place)
            else:
                o_pos, o_vel, o_w, o_ids, _ = orc.update_bvh(o_pos, o_vel, o_w, nsteps=k, nthreads=16, ids=o_ids,
                                                                 mode=orc.CONSISTENT if consistent else orc.AS_WRITTEN)
        except RuntimeError:
            cpu_degenerate = True
        cpu_s += time.perf_counter() - t0
        if gpu_degenerate or cpu_degenerate:
            out["degenerate_between_steps"] = [done, done + k]
            out["degenerate"] = {"device": gpu_degenerate, "oracle": cpu_degenerate}
            print(f"steps {done}..{done + k}: the tree became degenerate (depth cap) — device: {gpu_degenerate}, oracle: {cpu_degenerate}", flush=True)
            steps = done
            if gpu_degenerate != cpu_degenerate:
                out["checks"].append({"step": done + k, "bit_identical": False})
            break
        done += k
        p, v, w2, i2 = c.download()
        same = bool(np.array_equal(p.view(np.uint8), o_pos.view(np.uint8)) and np.array_equal(v.view(np.uint8), o_vel.view(np.uint8))
                    and np.array_equal(w2, o_w) and np.array_equal(i2, o_ids))
        out["checks"].append({"step": done, "bit_identical": same, "device_build": bool(c.last_build_on_device())})
        print(f"step {done}: bit identical to the oracle: {same}", flush=True)
        if not same:
            break
out["all_bit_identical"] = all(x["bit_identical"] for x in out["checks"]) and done == steps
out["steps_compared"] = done
out["gpu_ms_per_step"] = 1e3 * gpu_s / max(done, 1)
out["oracle_ms_per_step_16_threads"] = 1e3 * cpu_s / max(done, 1)
print(json.dumps(out))
sys.exit(0 if out["all_bit_identical"] else 1)
