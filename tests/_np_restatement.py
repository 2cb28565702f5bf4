"""Independent numpy restatement of the force law (src/main.rs:234-253), vectorised over sources.

A second, separately written reading of the reference used only to cross-check the C++ oracle (two
restatements that agree bit for bit make a transcription slip in either unlikely).  Test infrastructure.
"""
import numpy as np


def pair_terms(p, src, weight, clamp=0.001, dtype=np.float32):
    """Per-source contribution to the acceleration of target p: array [n_src, 2] in `dtype`, zeros where
    the reference returns early."""
    dt = np.dtype(dtype).type
    src = np.asarray(src, dtype=dtype).reshape(-1, 2)
    p = np.asarray(p, dtype=dtype)
    with np.errstate(all="ignore"):
        diff = src - p                                           # :236
        s = np.abs(diff[:, 0]) + np.abs(diff[:, 1])              # :238
        tiny = np.finfo(dtype).tiny
        normal = np.isfinite(s) & (np.abs(s) >= tiny)            # f32::is_normal  :241
        dist = diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]  # :245
        dist = np.where(dist < dt(clamp), dt(clamp), dist)       # :247-249
        force = np.asarray(weight, dtype=np.uint32).astype(dtype)
        num = diff * force[:, None]                              # diff * force
        den = (s * dist)[:, None]                                # sum * distance
        out = num / den                                          # :252
    out = np.where(normal[:, None], out, dt(0))
    return out.astype(dtype)


def direct_accel_seq(pos, weight, targets, clamp=0.001, dtype=np.float32):
    """Sequential ascending-j accumulation in `dtype` (python loop: small inputs only)."""
    pos = np.asarray(pos, dtype=dtype).reshape(-1, 2)
    out = np.zeros((len(targets), 2), dtype)
    for k, t in enumerate(targets):
        terms = pair_terms(pos[t], pos, weight, clamp, dtype)
        # a skipped pair leaves the accumulator untouched; adding +0 to an accumulator that started at +0 is the same
        ax = dtype(0)
        ay = dtype(0)
        for j in range(pos.shape[0]):
            ax = dtype(ax + terms[j, 0])
            ay = dtype(ay + terms[j, 1])
        out[k] = (ax, ay)
    return out
