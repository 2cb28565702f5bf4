"""Delta snapshots: stream size and hand-off time per snapshot against the plain snapshot (pos+vel+weight+ids)."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb


def run(name, pos, vel, w, method, every, rounds=8):
    world = nb.World(pos, vel, w, method=method)
    dec = nb.DeltaDecoder()
    cnt = nb.Counting()
    n = pos.shape[0]
    rows = []
    for k in range(rounds):
        if k:
            world.update(0.1, cnt, n_steps=every)
        t0 = time.perf_counter(); world.delta_begin(); t1 = time.perf_counter(); s, _ = world.delta_end(); t2 = time.perf_counter()
        dec.apply(s); t3 = time.perf_counter()
        world.snapshot_begin(); t4 = time.perf_counter(); world.snapshot_end(); t5 = time.perf_counter()
        rows.append((len(s), t1 - t0, t2 - t1, t3 - t2, t5 - t3))
    world.close()
    print(f"{name}: n={n} method={method} snapshot every {every} step(s); raw positions {8 * n} B, plain snapshot {20 * n} B")
    for k, (b, tb, te, td, ts) in enumerate(rows):
        print(f"  #{k}: stream {b:9d} B = {b / n:5.2f} B/body ({100 * b / (8 * n):5.1f} % of raw positions)  begin {1e3 * tb:6.3f} ms  end {1e3 * te:6.3f} ms"
              f"  host decode {1e3 * td:7.2f} ms   | plain snapshot begin+end {1e3 * ts:6.3f} ms")


if __name__ == "__main__":
    p, v, w = nb.scenes.galaxy()
    run("reference scene", p, v, w, "bvh", 1)
    run("reference scene", p, v, w, "bvh", 10)
    p, v, w = nb.scenes.plummer(1 << 20, seed=0x5EED0002, dtype=np.float32)
    run("Plummer 1M", p, v, w, "bvh", 1, rounds=5)
