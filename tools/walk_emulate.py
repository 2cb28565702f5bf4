"""CPU emulation of the wave-uniform BVH walk's control flow (no GPU): for a sample of waves of 64 tree-ordered targets,
how many node steps and leaf steps a wave takes and how many of its lanes act at each leaf step (the `takers`).
    python tools/walk_emulate.py [plummer|galaxy] [n] [waves]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402

C = nb._capi
scene = sys.argv[1] if len(sys.argv) > 1 else "plummer"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
n_waves = int(sys.argv[3]) if len(sys.argv) > 3 else 200
theta = np.float32(50.0)
if scene == "plummer":
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
else:
    pos, vel, w = nb.scenes.galaxy()
    n = pos.shape[0]
t = C.host_tree(C.TREE_BVH, pos, w)
geom, is_leaf, first, count, skip, order = t["geom"], t["is_leaf"], t["first"], t["count"], t["skip"], t["order"]
P = pos[order]                       # tree order
lo = geom[:, 0:2]
size = geom[:, 2:4]
hi = (lo + size).astype(np.float32)
cog = geom[:, 4:6]
s2 = (np.maximum(size[:, 0], size[:, 1]) ** 2).astype(np.float32)
m = len(is_leaf)
rng = np.random.default_rng(1)
starts = rng.choice(n // 64, n_waves, replace=False) * 64
hist = np.zeros(65, np.int64)
tot_node, tot_leaf, tot_rounds, tot_pairs_A, tot_acc = 0, 0, 0, 0, 0
per_wave = []
for s0 in starts:
    p = P[s0:s0 + 64]
    resume = np.zeros(len(p), np.int64)
    i = 0
    node_steps = leaf_steps = rounds = 0
    while i < m:
        act = resume <= i
        if is_leaf[i]:
            k = int(act.sum())
            if k:
                leaf_steps += 1
                hist[k] += 1
                rounds += k
            resume[act] = skip[i]
            i = skip[i]
        else:
            node_steps += 1
            contains = (p[:, 1] > lo[i, 1]) & (p[:, 0] > lo[i, 0]) & (p[:, 0] < hi[i, 0]) & (p[:, 1] < hi[i, 1])
            d = p - cog[i]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
            accept = act & ~contains & (s2[i] < d2 * theta * theta)
            desc = act & ~accept
            tot_acc += int(accept.sum())
            resume[accept] = skip[i]
            resume[desc] = i + 1
            i = i + 1 if desc.any() else skip[i]
    tot_node += node_steps
    tot_leaf += leaf_steps
    tot_rounds += rounds
    per_wave.append((node_steps, leaf_steps, rounds))
pw = np.array(per_wave)
print(f"{scene} n={n} nodes={m}: per wave (mean over {n_waves}): node steps {pw[:,0].mean():.0f} (max {pw[:,0].max()}), leaf steps {pw[:,1].mean():.0f} "
      f"(max {pw[:,1].max()}), (target, leaf) rounds {pw[:,2].mean():.0f}; accepted per target {tot_acc / (64 * n_waves):.1f}")
tk = np.arange(65)
print("takers per leaf step: mean %.2f; share of leaf steps with 1: %.2f, 2: %.2f, 3-4: %.2f, 5-8: %.2f, 9-32: %.2f, >32: %.2f" % (
    (hist * tk).sum() / hist.sum(), hist[1] / hist.sum(), hist[2] / hist.sum(), hist[3:5].sum() / hist.sum(), hist[5:9].sum() / hist.sum(),
    hist[9:33].sum() / hist.sum(), hist[33:].sum() / hist.sum()))
print("share of ROUNDS in leaf steps with 1: %.2f, 2: %.2f, 3-4: %.2f, 5-8: %.2f, 9-32: %.2f, >32: %.2f" % tuple(
    (hist * tk)[a:b].sum() / (hist * tk).sum() for a, b in ((1, 2), (2, 3), (3, 5), (5, 9), (9, 33), (33, 65))))
