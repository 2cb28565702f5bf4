"""Golden vectors of the delta-snapshot stream "NBD1" (csrc/delta_codec.h): frames and the streams the format's numpy
statement (oracle/delta_codec.py) encodes them to.  Pins the format: a later change to either side shows up here.
Run from the repo root: python tests/golden/make_golden_delta.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import delta_codec as dc  # noqa: E402


def frames(dtype, n, steps, seed):
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 2)) * 1e5).astype(dtype)
    vel = rng.standard_normal((n, 2)).astype(dtype)
    out = []
    for k in range(steps):
        out.append(pos.copy())
        vel = (vel + rng.standard_normal((n, 2)) * 0.01).astype(dtype)
        pos = (pos + vel * dtype(0.1)).astype(dtype)
    out[1][3] = (-0.0, np.inf)          # special bit patterns travel too
    out[2][3] = (np.nan, -1.5)
    return out


if __name__ == "__main__":
    data = {}
    for name, dtype, n, steps, seed in (("f32", np.float32, 130, 4, 11), ("f64", np.float64, 70, 3, 12)):
        fr = frames(dtype, n, steps, seed)
        enc = dc.Encoder()
        for k, f in enumerate(fr):
            data[f"{name}_frame{k}"] = f
            data[f"{name}_stream{k}"] = np.frombuffer(enc.encode(f, step=7 * k), np.uint8)
        data[f"{name}_steps"] = np.array(steps)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "delta_nbd1.npz"), **data)
    print({k: v.shape for k, v in data.items()})
