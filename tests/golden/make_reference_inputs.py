"""Inputs for tools/ref_golden.rs (the dumper a maintainer with `cargo` runs inside the reference to pin the oracle): small
seeded scenes as raw little-endian arrays under tests/golden/reference_inputs/<case>/.  Run from the repository root:
    python tests/golden/make_reference_inputs.py
The dumper's outputs go to tests/golden/from_reference/<case>/ and are consumed by tests/test_from_reference.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nbody_simulation_amd as nb  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference_inputs")


def write(case, pos, vel, w, steps):
    d = os.path.join(OUT, case)
    os.makedirs(d, exist_ok=True)
    np.ascontiguousarray(pos, "<f4").tofile(os.path.join(d, "pos0.f32"))
    np.ascontiguousarray(vel, "<f4").tofile(os.path.join(d, "vel0.f32"))
    np.ascontiguousarray(w, "<u4").tofile(os.path.join(d, "weight.u32"))
    with open(os.path.join(d, "steps.txt"), "w") as f:
        f.write(" ".join(str(s) for s in steps) + "\n")


def main():
    # config 1 of BASELINE.json: 1 024 bodies, 100 steps (the Plummer scene of tests/golden/config1_1024.npz)
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    write("a_plummer1024", pos, vel, w, (1, 10, 100))
    # the reference's own kind of scene: two heavy bodies, a thinned lattice, a disc — a subset that keeps the heavy ones
    pos, vel, w = nb.scenes.galaxy()
    sel = np.concatenate([[0, 1], np.arange(2, pos.shape[0], 30)])
    write("b_galaxy_subset", pos[sel], vel[sel], w[sel], (1, 5))
    # what pins the third-party semantics: coordinates of both signs and all-negative boxes (the max fold starts
    # from 0.0, bvh_tree.rs:42/59), a half-integer lattice (ties in nearly every add of the sequential sum; which of two
    # misplaced particles `partition` 0.1.2 swaps first decides the order inside a side, hence the next level's sum), masses
    # beyond 2^24 whose u32 sum wraps
    rng = np.random.default_rng(20261004)
    pos = np.concatenate([(rng.standard_normal((1500, 2)) * 3e4), (rng.integers(-400, 400, (1500, 2)) * 0.5)]).astype(np.float32)
    pos = pos + (np.arange(pos.shape[0])[:, None] * np.float32(1e-3)).astype(np.float32) * (np.arange(pos.shape[0])[:, None] % 7 == 0)
    vel = (rng.standard_normal(pos.shape) * 3).astype(np.float32)
    w = rng.integers(1, 9, pos.shape[0]).astype(np.uint32)
    w[::97] = 0x7FFFFFFF
    write("c_signs_ties_wrap", pos, vel, w, (1, 3))


if __name__ == "__main__":
    main()
