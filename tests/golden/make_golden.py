"""Generates tests/golden/*.npz from the CPU oracle (the reference is Rust and cannot run here, and it has no
fixtures of its own: SURVEY §4, §8c).  Committed together with its output so the vectors can be re-derived.

    python tests/golden/make_golden.py

Config 1 of BASELINE.json: 1 024 bodies, 100 steps.  Initial conditions: scenes.plummer(1024, seed 0x5EED0001).
Snapshots at steps 1, 10 and 100 (step 0 = the stored initial conditions) for
  bvh_as_written_theta50   World::update exactly as the reference runs it (theta 50, leaf 64, SURVEY F6 order)
  bvh_consistent_theta50   same with accelerations applied to the particle they were computed for
  bvh_as_written_theta0p5 / bvh_consistent_theta0p5
  quad_theta0p5            quad_tree.rs build + upward pass, walker by analogy, root cell (0,0,100000)
  direct                   the O(N^2) sum (theta = 0 limit), sequential f32 ascending j
plus known-answer vectors of the force law and one flattened BVH and quad tree.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
import nbody_simulation_amd as nb  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SNAPS = (1, 10, 100)


def main():
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    data = {"ic_pos": pos, "ic_vel": vel, "ic_weight": w}
    for name, theta, mode in (("bvh_as_written_theta50", 50.0, orc.AS_WRITTEN), ("bvh_consistent_theta50", 50.0, orc.CONSISTENT),
                              ("bvh_as_written_theta0p5", 0.5, orc.AS_WRITTEN), ("bvh_consistent_theta0p5", 0.5, orc.CONSISTENT)):
        p, v, ww, ids = pos, vel, w, None
        done = 0
        for s in SNAPS:
            p, v, ww, ids, _ = orc.update_bvh(p, v, ww, delta=0.1, theta=theta, mode=mode, nsteps=s - done, ids=ids)
            done = s
            data[f"{name}_s{s}_pos"], data[f"{name}_s{s}_vel"], data[f"{name}_s{s}_ids"] = p, v, ids
    p, v, done = pos, vel, 0
    for s in SNAPS:
        p, v, _ = orc.update_quad(p, v, w, delta=0.1, theta=0.5, nsteps=s - done)
        done = s
        data[f"quad_theta0p5_s{s}_pos"], data[f"quad_theta0p5_s{s}_vel"] = p, v
    p, v, done = pos, vel, 0
    for s in SNAPS:
        p, v, _ = orc.update_direct(p, v, w, delta=0.1, nsteps=s - done)
        done = s
        data[f"direct_s{s}_pos"], data[f"direct_s{s}_vel"] = p, v
    # accelerations at step 0
    data["direct_acc0"] = orc.direct_accel(pos, w)[0].astype(np.float32)
    bvh = orc.BVH(pos, w)
    data["bvh_theta0p5_acc0"] = bvh.walk(pos, theta=0.5)
    data["bvh_theta50_acc0"] = bvh.walk(pos, theta=50.0)
    t = bvh.flat()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "ids"):
        data[f"bvh_tree_{k}"] = getattr(t, k)
    q = orc.Quad(pos, w)
    data["quad_theta0p5_acc0"] = q.walk(pos, theta=0.5)
    tq = q.flat()
    for k in ("geom", "mass", "is_leaf", "depth", "child_code", "path", "first", "count", "skip", "order"):
        data[f"quad_tree_{k}"] = getattr(tq, k)
    # force-law known answers (p1, p2, mass) -> acc, including the skip cases
    tiny = float(np.finfo(np.float32).tiny)
    kat_in = np.array([[0, 0, 3, 4, 2], [0, 0, 1, 0, 1], [0, 0, -3, -4, 1], [10, 20, 7, 24, 750000], [0, 0, 0.01, 0, 1],
                       [5, 5, 5, 5, 9], [0, 0, 1e-39, 0, 1], [0, 0, tiny, 0, 1], [0, 0, np.inf, 0, 1], [0, 0, np.nan, 1, 1],
                       [0, 0, 3e38, 3e38, 1], [50000.5, 49999.25, 50001.75, 50003.5, 75000000]], np.float64)
    kat_out = np.array([orc.pair((r[0], r[1]), (r[2], r[3]), r[4]) for r in kat_in], np.float32)
    data["kat_in"], data["kat_out"] = kat_in, kat_out
    np.savez_compressed(os.path.join(OUT, "config1_1024.npz"), **data)
    print("wrote", os.path.join(OUT, "config1_1024.npz"), os.path.getsize(os.path.join(OUT, "config1_1024.npz")), "bytes")


if __name__ == "__main__":
    main()
