"""Per-step wall time of nbody_update_direct_f32 at small N: eager launches vs hipGraph replay of step pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import numpy as np, nbody_simulation_amd as nb
C = nb._capi
for n in (1024, 16384, 32768, 65536, 131072, 262144):
    pos, vel, w = nb.scenes.plummer(n, seed=3)
    row = []
    for graph in ("0", "1"):
        os.environ["NBODY_DIRECT_NEARFAR"] = "2" if graph == "1" else "0"
        with C.Context(0) as ctx:
            ctx.upload(pos, vel, w)
            ctx.update_direct(0.1, 10)
            steps = 400 if n <= 16384 else 100
            t0 = time.perf_counter(); ctx.update_direct(0.1, steps); wall = (time.perf_counter() - t0) / steps * 1e3
            p, v, _, _ = ctx.download()
            row.append((wall, p))
    same = bool(np.array_equal(row[0][1], row[1][1]))
    print(f"n={n}: nearfar off {row[0][0]:.3f} ms/step, on {row[1][0]:.3f} ms/step, x{row[0][0]/row[1][0]:.2f}, identical results: {same}", flush=True)
