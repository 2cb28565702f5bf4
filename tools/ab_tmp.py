import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nbody_simulation_amd as nb
C = nb._capi
for n in (1 << 20, 200003, 65536):
    pos, vel, w = nb.scenes.plummer(n, seed=11)
    out = {}
    for m in ("2", "3"):
        os.environ["NBODY_DIRECT_ASM"] = m
        with C.Context(0) as ctx:
            ctx.set_params(arith=C.ARITH_AUTO)
            ctx.upload(pos, vel, w)
            out[m] = ctx.accel_direct().copy()
    print(n, "bitwise equal 2 vs 3:", np.array_equal(out["2"], out["3"]), float(np.abs(out["2"] - out["3"]).max()), flush=True)
    w5 = (1 + (np.arange(n) % 5)).astype(np.uint32)
    for m in ("2", "3"):
        os.environ["NBODY_DIRECT_ASM"] = m
        with C.Context(0) as ctx:
            ctx.set_params(arith=C.ARITH_AUTO)
            ctx.upload(pos, vel, w5)
            out[m] = ctx.accel_direct().copy()
    print(n, "classes: bitwise equal 2 vs 3:", np.array_equal(out["2"], out["3"]), float(np.abs(out["2"] - out["3"]).max()), flush=True)
