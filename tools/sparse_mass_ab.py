"""A/B of the sparse-heavy-mass path of the direct step: scenes whose masses are all equal but for a few bodies
(the reference's own scene: 2 heavy among 151 k, main.rs:282-291) with the odd bodies riding with the near list
(default) against the per-body-mass kernel (NBODY_DIRECT_NO_SPARSE=1).  Prints one JSON line per case.
    python tools/sparse_mass_ab.py            (needs an MI355X)
"""
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


def run(name, pos, vel, w, steps):
    out = {"case": name, "n": int(pos.shape[0])}
    for label, env in (("sparse", "0"), ("per_body", "1")):
        os.environ["NBODY_DIRECT_NO_SPARSE"] = env
        timer = C.Timer()
        with C.Context(0) as c:
            c.upload(pos, vel, w)
            c.update_direct(0.1, 1)
            c.set_timer(timer)
            timer.read(reset=True)
            t0 = time.perf_counter()
            c.update_direct(0.1, steps)
            dt = time.perf_counter() - t0
            kms, kl = timer.read(reset=True)
            c.set_timer(None)
        n = float(pos.shape[0])
        out[label] = {"ms_per_step": 1e3 * dt / steps, "main_kernel_ms": kms, "frac_of_f32_peak": 14 * n * n / (dt / steps) / 157.3e12}
    print(json.dumps(out), flush=True)


def main():
    pos, vel, w = nb.scenes.galaxy()
    run("reference scene (World::new): 2 heavy bodies", pos, vel, w, 20)
    n = 1 << 20
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
    run("Plummer 1M, equal masses (the headline input)", pos, vel, w, 3)
    w2 = w.copy()
    w2[np.arange(0, n, n // 100)[:100]] = 1_000_000
    run("Plummer 1M, 100 heavy bodies", pos, vel, w2, 3)
    w3 = (np.arange(n) % 5 + 1).astype(np.uint32)
    run("Plummer 1M, per-body masses 1..5", pos, vel, w3, 3)


if __name__ == "__main__":
    main()
