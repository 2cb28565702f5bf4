"""NBODY_WALK_WAVE_LOG=1: per-wave clock ticks (100 MHz) and step counts of the FAST one-pass BVH walk (walk_tile_fast).
    python tools/walk_wave_log.py [galaxy|plummer] [steps before the logged one = 2]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NBODY_WALK_WAVE_LOG"] = "1"
os.environ["NBODY_STEP_AHEAD"] = "0"
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb
C = nb._capi
scene = sys.argv[1] if len(sys.argv) > 1 else "galaxy"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pos, vel, w = nb.scenes.galaxy() if scene == "galaxy" else nb.scenes.plummer(1 << 20, seed=0x5EED0003)
with C.Context(0) as ctx:
    ctx.set_params(theta=50.0, order=C.ORDER_AS_WRITTEN, arith=C.ARITH_FAST)
    ctx.upload(pos, vel, w)
    t = C.Timer()
    os.environ["NBODY_WALK_WAVE_LOG"] = "0"
    ctx.update_tree(C.TREE_BVH, 0.1, warm)
    os.environ["NBODY_WALK_WAVE_LOG"] = "1"
    ctx.set_timer(t)
    ctx.update_tree(C.TREE_BVH, 0.1, 1)
    ms, _ = t.read()
log = np.fromfile("/tmp/nbody_wave_log.bin", dtype=np.uint64).reshape(-1, 4)
start = (log[:, 0] >> np.uint64(24)).astype(np.int64)
us = (log[:, 0] & np.uint64(0xFFFFFF)) * 0.01
nodes, leaves = (log[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64), (log[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
t_search = ((log[:, 1] >> np.uint64(32)) & np.uint64(0xFFFF)) * 0.01    # us from the wave's start to the end of its target search
t_first = ((log[:, 1] >> np.uint64(48)) & np.uint64(0xFFFF)) * 0.01     # ... to the arrival of the root's record and the targets
t_tail = ((log[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)) * 0.01      # us of the epilogue (history, total, the log itself)
targets, rounds = (log[:, 3] >> np.uint64(32)).astype(np.int64), (log[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
live = targets > 0
print(f"{scene} after {warm} steps, NBODY_WALK_TILE_WAVES={os.environ.get('NBODY_WALK_TILE_WAVES', 'default')}: walk kernel {ms:.3f} ms (with the log's overhead); {live.sum()} of {len(us)} waves have targets")
print(f"per live wave: {us[live].mean():.1f} us mean, {np.percentile(us[live], 99):.1f} p99, {us[live].max():.1f} max; steps {nodes[live].mean():.0f} node + "
      f"{leaves[live].mean():.0f} leaf, {rounds[live].mean():.0f} rounds, {targets[live].mean():.1f} targets")
print(f"a wave's fixed part: search {t_search[live].mean():.1f} us mean ({np.percentile(t_search[live], 99):.1f} p99), root record and targets there after "
      f"{t_first[live].mean():.1f} us ({np.percentile(t_first[live], 99):.1f} p99), epilogue {t_tail[live].mean():.1f} us")
steps = nodes + leaves
ok = live & (steps > 0)
print(f"us per step (node + leaf): mean {(us[ok] / steps[ok]).mean():.2f}; sum of wave times {us[live].sum() / 1e3:.1f} ms")
A = np.stack([nodes[ok], leaves[ok], rounds[ok], np.ones(ok.sum())], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, us[ok], rcond=None)
print("least squares: %.3f us per node step + %.3f us per leaf step + %.3f us per round + %.1f us per wave" % tuple(coef))
t_begin = start[live].min()
end = start + (us * 100).astype(np.int64)
print(f"waves start between 0 and {(start[live].max() - t_begin) * 0.01:.1f} us after the first; the last one ends at {(end[live].max() - t_begin) * 0.01:.1f} us")
for k in np.argsort(-end * live)[:5]:
    print(f"  ends last: wave {k}: starts at {(start[k] - t_begin) * 0.01:.1f} us, runs {us[k]:.1f} us, {targets[k]} targets, {nodes[k]} node steps, {leaves[k]} leaf steps")
h = np.histogram(us[live], bins=[0, 25, 50, 75, 100, 125, 150, 175, 200, 250, 300, 400, 1000])[0]
print("waves by run time (us) <25 <50 <75 <100 <125 <150 <175 <200 <250 <300 <400 more:", " ".join(str(x) for x in h))
big = live & (targets >= 48)
few = live & (targets <= 8)
for name, m in (("waves of >= 48 targets", big), ("waves of <= 8 targets", few)):
    if m.sum():
        print(f"  {name}: {m.sum()}, {us[m].mean():.1f} us mean, {us[m].max():.1f} max, {nodes[m].mean():.0f} node + {leaves[m].mean():.0f} leaf steps, {rounds[m].mean():.0f} rounds")
for k in np.argsort(-us)[:8]:
    print(f"  wave {k}: {us[k]:.1f} us, {targets[k]} targets, {nodes[k]} node steps, {leaves[k]} leaf steps, {rounds[k]} rounds")
