"""Round 4: how many waves should the one-pass BVH walk aim at?  Whole World::update steps (BVH, theta 50, AS_WRITTEN) on the
reference scene and on Plummer spheres of three sizes, for several NBODY_WALK_TILE_WAVES (laboratory library) and the product's
size-aware default (tile_waves_target, csrc/walk_split.hip).  python tools/walk_wave_target.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")
import nbody_simulation_amd as nb  # noqa: E402
C = nb._capi


def run(pos, vel, w, steps, arith):
    timer = C.Timer()
    with C.Context(0) as c:
        c.set_params(theta=50.0, order=C.ORDER_AS_WRITTEN, arith=arith)
        c.upload(pos, vel, w)
        c.update_tree(C.TREE_BVH, 0.1, 2)
        c.set_timer(timer)
        t0 = time.perf_counter()
        c.update_tree(C.TREE_BVH, 0.1, steps)
        dt = time.perf_counter() - t0
        kms, _ = timer.read()
        c.set_timer(None)
    return 1e3 * dt / steps, kms


# python tools/walk_wave_target.py [all | ref | plummer] [W ...]   (default: all, the values below)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
f64 = which.endswith("64")                              # "plummer64": the same spheres in f64 (walk_tile<double>)
which = which[:-2] if f64 else which
values = tuple(sys.argv[2:]) or ("default", "16384", "8192", "6144")
scenes = [("reference scene", nb.scenes.galaxy(), 300)] if which in ("all", "ref") else []
if which in ("all", "plummer"):
    for n, steps in ((262144, 100), (400000, 60), (655360, 40), (1 << 20, 20)) + (((1 << 21, 12),) if os.environ.get("WALK_SWEEP_2M") else ()):
        scenes.append((f"plummer {n}" + (" f64" if f64 else ""), nb.scenes.plummer(n, seed=0x5EED0003, dtype=np.float64 if f64 else np.float32),
                       steps // 4 if f64 else steps))
print(f"{'scene':<18} {'NBODY_WALK_TILE_WAVES':<22} {'exact ms/step (kernel)':<26} FAST ms/step (kernel)")
for name, (pos, vel, w), steps in scenes:
    for waves in values:
        if waves == "default":
            os.environ.pop("NBODY_WALK_TILE_WAVES", None)
        elif waves.startswith("+"):                           # "+k": k waves more than the head count n / 64
            os.environ["NBODY_WALK_TILE_WAVES"] = str(pos.shape[0] // 64 + int(waves[1:]))
        else:
            os.environ["NBODY_WALK_TILE_WAVES"] = waves
        e = run(pos, vel, w, steps, C.ARITH_AUTO)
        f = run(pos, vel, w, steps, C.ARITH_FAST)
        print(f"{name:<18} {waves:<22} {e[0]:7.3f} ({e[1]:6.3f})          {f[0]:7.3f} ({f[1]:6.3f})", flush=True)
