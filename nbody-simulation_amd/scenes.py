"""Seeded initial conditions (the reference's own are unseeded and irreproducible).

plummer():   2-D projection of a Plummer sphere — the synthetic input of every BASELINE.json config.
             Counter-based RNG (splitmix64 of seed and a per-body counter), so any slice of the bodies can
             be generated independently (each rank of a sharded run generates the same full set).
galaxy():    the scene of World::new (main.rs:276-346) with a seeded generator: two heavy bodies, a thinned
             lattice disc around the second, 100 000 bodies in a uniform-angle/uniform-radius disc.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """One splitmix64 output per input word (uint64 arithmetic wraps)."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _uniform(seed: int, index: np.ndarray, stream: int) -> np.ndarray:
    """Uniform in (0,1): 53 random bits of splitmix64(splitmix64(seed ^ stream-tag) + index), centred."""
    with np.errstate(over="ignore"):
        key = splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))],
                                  dtype=np.uint64))[0]
        bits = splitmix64((index.astype(np.uint64) + key) & _M64)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def plummer(n: int, seed: int = 0x5EED0000, *, scale: float = 5000.0, clip: float = 45000.0,
            centre=(50000.0, 50000.0), vscale: float = 1.0, dtype=np.float32, start: int = 0):
    """Bodies start..start+n of the seeded Plummer set.  -> (pos[n,2], vel[n,2], weight[n] u32 == 1).

    r = a / sqrt(u^(-2/3) - 1) clipped to `clip`, isotropic 3-D direction with z dropped, centred so every
    coordinate lies in (0, 100000) (HEIGHT, main.rs:31).  Speeds by Aarseth-Henon-Wielen rejection sampling,
    scaled so that |v| <= vscale (the reference's random bodies have |v| <= 1, main.rs:260-268)."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    u = _uniform(seed, idx, 1)
    r = scale / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
    r = np.minimum(r, clip)
    z = 2.0 * _uniform(seed, idx, 2) - 1.0
    phi = 2.0 * np.pi * _uniform(seed, idx, 3)
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    x = centre[0] + r * s * np.cos(phi)
    y = centre[1] + r * s * np.sin(phi)
    # speed: q in (0,1) with density q^2 (1-q^2)^3.5, by rejection over a fixed number of rounds
    q = np.full(n, 0.5)
    todo = np.ones(n, bool)
    for k in range(64):
        if not todo.any():
            break
        a = _uniform(seed, idx, 10 + 2 * k)
        b = _uniform(seed, idx, 11 + 2 * k)
        ok = todo & (0.1 * b < a * a * (1.0 - a * a) ** 3.5)
        q[ok] = a[ok]
        todo &= ~ok
    vesc = np.sqrt(2.0) * (1.0 + (r / scale) ** 2) ** -0.25
    speed = vscale * q * vesc / np.sqrt(2.0)
    vz = 2.0 * _uniform(seed, idx, 4) - 1.0
    vphi = 2.0 * np.pi * _uniform(seed, idx, 5)
    vs = np.sqrt(np.maximum(0.0, 1.0 - vz * vz))
    vx = speed * vs * np.cos(vphi)
    vy = speed * vs * np.sin(vphi)
    pos = np.stack([x, y], axis=1).astype(dtype)
    vel = np.stack([vx, vy], axis=1).astype(dtype)
    return pos, vel, np.ones(n, np.uint32)


def free_weights(n: int, seed: int = 0x5EED0003, top: int = 100_000, start: int = 0) -> np.ndarray:
    """Free per-body masses: u32 weights 1 .. `top` from splitmix64(seed ^ index) — far more than 32 distinct values, so
    neither the equal-mass hoist nor the mass classes of the direct step apply (main.rs:193-198: every particle carries its
    own `weight: u32`, used as `weight as f32`, :360).  bench.py's `free_masses` leg and its parity test use these."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = splitmix64(idx ^ np.uint64(seed))
    return (z % np.uint64(top) + np.uint64(1)).astype(np.uint32)


def galaxy(seed: int = 0xC0FFEE, dtype=np.float32):
    """The scene of World::new (main.rs:276-346), seeded.  N ~ 151 000 (2 + ~51 k lattice + 100 000)."""
    height = 100_000
    circle1 = np.array([35000.0, 35000.0], np.float32)
    circle2 = np.array([60000.0, 60000.0], np.float32)
    c1lenr2 = np.float32(15000000.0)
    pos = [circle1, circle2]                                   # :282-291
    vel = [np.array([200.0, 250.0], np.float32), np.zeros(2, np.float32)]
    w = [75_000_000, 750_000]
    m = height // 14 - 1                                       # :316-317
    gx, gy = np.meshgrid(np.arange(m, dtype=np.float32) * np.float32(14.0),
                         np.arange(m, dtype=np.float32) * np.float32(14.0), indexing="ij")
    p = np.stack([gx.ravel(), gy.ravel()], axis=1)             # x outer, y inner, as the loops nest
    d = p - circle2
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
    u = _uniform(seed, np.arange(p.shape[0], dtype=np.uint64), 1).astype(np.float32)
    draw = u * (c1lenr2 - d2 + np.float32(1.0))                # gen_range(0..(c1lenr2 - d2) + 1.0)  :321
    keep = (d2 < c1lenr2) & (d2 > np.float32(500000.0)) & (draw > np.float32(6000000.0))
    p, d, d2 = p[keep], d[keep], d2[keep]
    scale = np.sqrt(np.sqrt(np.float32(750000.0)) / d2).astype(np.float32)   # :323-324
    v = np.stack([d[:, 1], -d[:, 0]], axis=1) * scale[:, None]               # rotate_right :271-273
    n_disc = 100_000                                           # :334
    idx = np.arange(n_disc, dtype=np.uint64)
    offset = np.array([50000.0, 50000.0], np.float32)

    def rand_disc(s0):                                         # :255-258
        th = (_uniform(seed, idx, s0) * 2.0 * np.pi).astype(np.float32)
        rr = _uniform(seed, idx, s0 + 1).astype(np.float32)
        return np.stack([np.cos(th), np.sin(th)], axis=1).astype(np.float32) * rr[:, None]

    bp = rand_disc(20) * np.float32(25000.0) + offset          # :260-268
    bv = rand_disc(22)
    pos = np.concatenate([np.stack(pos), p, bp]).astype(dtype)
    vel = np.concatenate([np.stack(vel), v.astype(np.float32), bv]).astype(dtype)
    weight = np.concatenate([np.array(w, np.uint32), np.ones(p.shape[0] + n_disc, np.uint32)])
    return pos, vel, weight
