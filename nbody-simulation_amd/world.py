"""Host-side mirror of the reference's step interface, above the C ABI.

Reference (Rust, /root/reference src/main.rs):
    struct World { particles: Vec<Particle> }                       :37-39
    struct Counting { build_bvh, sum_gravity, post_calculations }   :74-79
    impl World { fn update(&mut self, delta: f32, counter: &mut Counting) }   :388-425
    called once per simulation step as `world.update(STEP_SIZE, &mut counter)`  :120

Same names and argument meaning here; the particle arrays live on the GPU between calls and are read back
with `particles()`.  `method` selects what the force phase is: "bvh" is the reference's own path, "quad" the
quad_tree.rs tree (dead code upstream), "direct" the O(N^2) sum (theta = 0 limit).
"""
from __future__ import annotations

import numpy as np

from . import _capi
from ._capi import Counting  # noqa: F401  (re-exported: the reference's struct name)

STEP_SIZE = 0.1   # main.rs:34
THETA = 50.0      # main.rs:35
HEIGHT = 100_000  # main.rs:31

_METHODS = {"bvh": _capi.TREE_BVH, "quad": _capi.TREE_QUAD, "direct": None}


class World:
    def __init__(self, position, velocity, weight=None, *, method="bvh", device=0, devices=None, theta=THETA, clamp=0.001,
                 leaf_size=64, order="as_written", arith="auto", quad_root=(0.0, 0.0, float(HEIGHT))):
        """`devices=[0, 1, ...]`: the same world on several GPUs of one node behind one handle (nbody_create_multi);
        `update` is still the one call of main.rs:120, the library shards the step and exchanges the results."""
        if method not in _METHODS:
            raise ValueError(f"method must be one of {sorted(_METHODS)}")
        self.method = method
        self.ctx = _capi.MultiContext(list(devices)) if devices is not None else _capi.Context(device)
        self.ctx.set_params(theta=float(theta), clamp=float(clamp), leaf_size=int(leaf_size),
                            order={"as_written": _capi.ORDER_AS_WRITTEN, "consistent": _capi.ORDER_CONSISTENT}[order],
                            arith={"auto": _capi.ARITH_AUTO, "fast": _capi.ARITH_FAST, "exact": _capi.ARITH_EXACT}[arith],
                            quad_root_x=float(quad_root[0]), quad_root_y=float(quad_root[1]),
                            quad_root_h=float(quad_root[2]))
        self.ctx.upload(position, velocity, weight)
        if method == "direct" and self.ctx.dtype != np.float32:
            raise ValueError("the direct path is f32 (the reference's precision)")

    def update(self, delta: float, counter: Counting | None = None, n_steps: int = 1):
        """World::update (main.rs:388-425): build, force, integrate; accumulates phase seconds into `counter`."""
        if self.method == "direct":
            self.ctx.update_direct(delta, n_steps, counter)
        else:
            self.ctx.update_tree(_METHODS[self.method], delta, n_steps, counter)

    def update_async(self, delta: float, n_steps: int = 1):
        """The same step(s) without waiting for them (f32 tree methods; nbody_update_tree_async_f32): the reference's loop
        — update, then hand a snapshot over if the channel has room, main.rs:118-139 — keeps the device busy from one
        iteration to the next.  `wait()` completes them and returns the cumulative Counting."""
        if self.method == "direct" or self.ctx.dtype != np.float32:
            self.update(delta, None, n_steps)
        else:
            self.ctx.update_tree_async(_METHODS[self.method], delta, n_steps)

    def wait(self) -> Counting:
        self.ctx.wait()
        return self.ctx.counting()

    def particles(self):
        """-> (position[n,2], velocity[n,2], weight[n], ids[n]); rows are in the order the reference's
        `self.particles` would be in (permuted by every BVH build); ids give each row's original index."""
        return self.ctx.download()

    def snapshot_begin(self):
        """Hand the current particles to a consumer without stopping the simulation (the `try_send` of
        main.rs:136-139): returns at once, `update` may be called right away."""
        self.ctx.snapshot_begin()

    def snapshot_end(self):
        """-> (position, velocity, weight, ids, updates) of the pending snapshot."""
        return self.ctx.snapshot_end()

    def delta_begin(self):
        """Like snapshot_begin, but hands over only what changed: the positions, in id order, as a lossless stream
        of the change against the previous delta snapshot (the idea of the commented experiment, main.rs:107-134)."""
        self.ctx.delta_begin()

    def delta_end(self):
        """-> (stream bytes, updates); feed the streams in order to a `DeltaDecoder`."""
        return self.ctx.delta_end()

    def frame(self, height=100_000, render_px=1250):
        """The frame the reference's render thread would draw from these particles (`draw`, main.rs:41-72):
        uint8 array (render_px, render_px, 4), RGBA."""
        return self.ctx.render(height, render_px)

    def save_frame(self, path, height=100_000, render_px=1250):
        """Write the frame as an RGBA PNG (what the reference shows in its window)."""
        write_png(path, self.frame(height, render_px))

    def close(self):
        self.ctx.close()


def write_png(path, rgba):
    """Minimal PNG writer (8-bit RGBA, zlib only)."""
    import struct
    import zlib
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, _ = rgba.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rgba.reshape(h, w * 4)], axis=1).tobytes()   # filter 0 per row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
