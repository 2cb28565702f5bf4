"""Generates tests/golden/frames.npz: frames of the reference's draw() (main.rs:41-72) from the CPU oracle's line-by-line
restatement (the reference has no fixtures and cannot run here: SURVEY §8c).  Committed with its output.

    python tests/golden/make_golden_frames.py

  galaxy_125        the seeded reference scene (scenes.galaxy()) as uploaded, RENDER_HEIGHT = 125 (cell = 800 units)
  galaxy_1250_nz    the same at the reference's 1250: indices and RGBA of the non-zero pixels only
  plummer_s100_125  config 1 after 100 BVH steps (as-written order), rows in the order the in-place partition left them
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
import nbody_simulation_amd as nb  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    data = {}
    pos, vel, w = nb.scenes.galaxy()
    data["galaxy_125"] = orc.draw(pos, vel, w, 100_000, 125)
    full = orc.draw(pos, vel, w, 100_000, 1250).reshape(-1, 4)
    nz = np.flatnonzero(full[:, 3])
    data["galaxy_1250_nz_index"] = nz.astype(np.uint32)
    data["galaxy_1250_nz_rgba"] = full[nz]
    g = np.load(os.path.join(OUT, "config1_1024.npz"))
    ids = g["bvh_as_written_theta50_s100_ids"]
    data["plummer_s100_125"] = orc.draw(g["bvh_as_written_theta50_s100_pos"], g["bvh_as_written_theta50_s100_vel"],
                                        g["ic_weight"][ids], 100_000, 125)
    np.savez_compressed(os.path.join(OUT, "frames.npz"), **data)
    print({k: v.shape for k, v in data.items()})


if __name__ == "__main__":
    main()
