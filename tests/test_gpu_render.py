"""The frame raster (render.hip) against the oracle's line-by-line draw() (main.rs:41-72).  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_render_equals_reference_draw_on_crowded_pixels(nb, orc, dtype):
    C = nb._capi
    rng = np.random.default_rng(31)
    n = 200_000
    pos = (rng.random((n, 2)) * 1.1e5 - 5e3).astype(dtype)                    # some rows outside [0, HEIGHT)^2
    pos[:50_000] = (rng.random((50_000, 2)) * 2000 + 40_000).astype(dtype)    # ~80 rows per pixel: alpha saturates
    pos[7] = (np.nan, 10)
    vel = (rng.standard_normal((n, 2)) * 4).astype(dtype)
    vel[11] = (np.inf, 0)
    vel[12] = (np.nan, 0)
    w = np.where(rng.random(n) < 0.001, 750_000, rng.integers(1, 12, n)).astype(np.uint32)   # weights 10 / 11 straddle "> 10"
    with C.Context(0) as ctx:
        ctx.upload(pos, vel, w)
        for px in (1250, 100):
            got = ctx.render(100_000, px)
            want = orc.draw(pos, vel, w, 100_000, px)
            assert np.array_equal(got, want), px
        with pytest.raises(C.NBodyError):
            ctx.render(100_000, 1251)          # does not divide HEIGHT: out-of-range index upstream


def test_render_follows_the_row_order_of_the_bvh_permutation(nb, orc):
    """draw() runs over `world.particles` as the in-place partition left them: the last light row on a pixel wins."""
    pos, vel, w = nb.scenes.galaxy()
    world = nb.World(pos, vel, w, method="bvh")
    try:
        cnt = nb.Counting()
        for _ in range(3):
            world.update(0.1, cnt)
        p, v, w2, _ = world.particles()
        frame = world.frame()
        assert np.array_equal(frame, orc.draw(p, v, w2))
        assert frame[..., 3].any() and (frame[..., 1] == 255).any()   # the two heavy bodies are green
    finally:
        world.close()


def test_render_empty_and_single(nb, orc):
    C = nb._capi
    with C.Context(0) as ctx:
        pos = np.array([[85, 170]], F32)
        vel = np.array([[0.3, -0.4]], F32)
        ctx.upload(pos, vel, np.ones(1, np.uint32))
        f = ctx.render()
        assert tuple(f[2, 1]) == (255, 232, 232, 10) and np.count_nonzero(f) == 4


# ------------------------------------------------------------------ snapshot hand-off (main.rs:136-139)
@pytest.mark.parametrize("method,dtype", [("bvh", np.float32), ("quad", np.float64), ("direct", np.float32)])
def test_snapshot_is_the_state_at_begin_even_if_steps_follow(nb, method, dtype):
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(20000, seed=8, dtype=dtype)
    w = (np.arange(20000) % 5 + 1).astype(np.uint32)
    world = nb.World(pos, vel, w, method=method)
    ref = nb.World(pos, vel, w, method=method)
    try:
        cnt = nb.Counting()
        for _ in range(2):
            world.update(0.1, cnt)
            ref.update(0.1, cnt)
        want = ref.particles()
        world.snapshot_begin()
        assert world.ctx.snapshot_pending()
        with pytest.raises(C.NBodyError):
            world.snapshot_begin()             # "channel full"
        for _ in range(3):
            world.update(0.1, cnt)             # overwrites the rows while the snapshot travels
        p, v, w2, ids, updates = world.snapshot_end()
        assert updates == 2 and not world.ctx.snapshot_pending()
        for a, b in zip((p, v, w2, ids), want):
            assert np.array_equal(a, b)
        with pytest.raises(C.NBodyError):
            world.snapshot_end()               # nothing pending
        for _ in range(3):
            ref.update(0.1, cnt)
        for a, b in zip(world.particles(), ref.particles()):   # the snapshot did not disturb the run
            assert np.array_equal(a, b)
    finally:
        world.close()
        ref.close()


def test_headless_driver_writes_the_frames_the_window_would_show(nb, orc, tmp_path):
    """nbody_run ... frame_every prefix: PAM files of draw() every k steps; the last one equals the oracle's draw() of a
    Python run of the same (seeded) scene would need the C++ generator, so check structure + content sanity here and
    the PNG writer against the frame it was given."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(nb._capi.LIB_PATH), "nbody_run")
    if not os.path.exists(exe):
        pytest.skip("nbody_run not built")
    prefix = str(tmp_path / "f")
    r = subprocess.run([exe, "4", "bvh", "777", "2", prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for step in (2, 4):
        raw = open(f"{prefix}_{step:06d}.pam", "rb").read()
        head, body = raw.split(b"ENDHDR\n", 1)
        assert head.startswith(b"P7\nWIDTH 1250\nHEIGHT 1250\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA")
        f = np.frombuffer(body, np.uint8).reshape(1250, 1250, 4)
        heavy = (f[..., 0] == 0) & (f[..., 1] == 255) & (f[..., 3] == 255)
        assert heavy.sum() == 2                                   # the two heavy bodies, green
        assert (f[..., 3] > 0).sum() > 20000                      # the disc and the cloud
    # Python side: PNG of a frame
    pos, vel, w = nb.scenes.plummer(5000, seed=3)
    world = nb.World(pos, vel, w, method="direct")
    try:
        path = str(tmp_path / "frame.png")
        world.save_frame(path)
        import struct
        import zlib
        d = open(path, "rb").read()
        i, idat = 8, b""
        while i < len(d):
            n = struct.unpack(">I", d[i:i + 4])[0]
            if d[i + 4:i + 8] == b"IDAT":
                idat += d[i + 8:i + 8 + n]
            i += 12 + n
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(1250, 1 + 1250 * 4)
        assert np.array_equal(rows[:, 1:].reshape(1250, 1250, 4), orc.draw(pos, vel, w))
    finally:
        world.close()
