"""What sharding costs besides the pairs: the headline step (N = 1 048 576, direct) through nbody_create_multi with the ONE
GPU listed 1, 2, 4, 8 times (peer-copy exchange).  The ranks share the device, so the pair work is the single context's;
what is added is G preparations (hazard scan + near/far split, replicated per rank), the block exchange and the workers'
barriers.  Not a scaling measurement.    python tools/multi_overhead.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi
n = 1 << 20
pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
with C.Context(0) as c:
    c.upload(pos, vel, w)
    c.update_direct(0.1, 1)
    t0 = time.perf_counter(); c.update_direct(0.1, 3); t1 = (time.perf_counter() - t0) / 3
print(json.dumps({"ranks": "single context", "ms_per_step": round(1e3 * t1, 3)}), flush=True)
for g in (1, 2, 4, 8):
    for chunks in ((0,) if g == 1 else (0, 2)):
        with C.MultiContext([0] * g, C.EXCHANGE_PEER, chunks) as m:
            m.upload(pos, vel, w)
            m.update_direct(0.1, 1)
            t0 = time.perf_counter(); m.update_direct(0.1, 3); t = (time.perf_counter() - t0) / 3
            info = m.multi_info()
        print(json.dumps({"ranks": g, "chunks": info[2], "block": info[3], "ms_per_step": round(1e3 * t, 3),
                          "over_single_ms": round(1e3 * (t - t1), 3)}), flush=True)
