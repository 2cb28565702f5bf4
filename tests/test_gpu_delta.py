"""Delta snapshots: the device encoder (delta_snapshot.hip) against the numpy statement of the format
(oracle/delta_codec.py), byte for byte, and the round trip through the host decoder.  Needs an MI355X."""
import numpy as np
import pytest

from oracle import delta_codec as dc

pytestmark = pytest.mark.gpu


def _by_id(world):
    p, _, _, ids = world.particles()
    out = np.empty_like(p)
    out[ids] = p
    return out


def _same_bits(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("method,dtype,n", [("direct", np.float32, 5000), ("bvh", np.float32, 20_000),
                                            ("quad", np.float64, 6001), ("quad", np.float32, 64), ("bvh", np.float64, 777)])
def test_device_stream_equals_the_format_statement(nb, method, dtype, n):
    """Uneven cadence on purpose (1, 1, 3, 1 steps apart): the predictor choice is per block and must still agree."""
    pos, vel, w = nb.scenes.plummer(n, seed=77, dtype=dtype)
    world = nb.World(pos, vel, w, method=method)
    enc, dec = dc.Encoder(), nb.DeltaDecoder()
    try:
        cnt = nb.Counting()
        done = 0
        sizes = []
        for gap in (0, 1, 1, 3, 1):
            if gap:
                world.update(0.1, cnt, n_steps=gap)
                done += gap
            world.delta_begin()
            assert world.ctx.delta_pending()
            stream, step = world.delta_end()
            assert not world.ctx.delta_pending()
            now = _by_id(world)
            assert step == done
            assert stream == enc.encode(now, step=done)
            dec.apply(stream)
            assert _same_bits(dec.positions(), now)
            sizes.append(len(stream))
        assert sizes[2] < sizes[0]            # a delta is smaller than the key frame
    finally:
        world.close()


def test_stream_is_the_state_at_begin_and_steps_may_follow(nb):
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    world = nb.World(pos, vel, w, method="bvh")
    dec = nb.DeltaDecoder()
    try:
        cnt = nb.Counting()
        world.update(0.1, cnt)
        world.delta_begin()
        at_begin = _by_id(world)
        with pytest.raises(C.NBodyError):
            world.delta_begin()                         # one in flight
        world.snapshot_begin()                          # the plain snapshot is independent of it
        world.update(0.1, cnt, n_steps=2)
        stream, step = world.delta_end()
        assert step == 1
        dec.apply(stream)
        assert _same_bits(dec.positions(), at_begin)
        p, _, _, ids, sstep = world.snapshot_end()
        byid = np.empty_like(p)
        byid[ids] = p
        assert sstep == 1 and _same_bits(byid, at_begin)
        world.delta_begin()                             # the next stream spans the two steps
        stream, step = world.delta_end()
        dec.apply(stream)
        assert step == 3 and _same_bits(dec.positions(), _by_id(world))
        with pytest.raises(C.NBodyError):
            world.delta_end()                           # nothing pending
    finally:
        world.close()


def test_small_buffer_reset_and_new_upload(nb):
    C = nb._capi
    rng = np.random.default_rng(4)
    n = 1000
    pos = (rng.random((n, 2)) * 1e5).astype(np.float32)
    vel = rng.standard_normal((n, 2)).astype(np.float32)
    with C.Context(0) as ctx:
        with pytest.raises(C.NBodyError):
            ctx.delta_begin()                           # nothing uploaded
        ctx.upload(pos, vel, np.ones(n, np.uint32))
        ctx.delta_begin()
        with pytest.raises(C.NBodyError, match="smaller than the stream"):
            ctx.delta_end(cap=64)
        assert ctx.delta_pending()                      # still there
        with pytest.raises(C.NBodyError):
            ctx.delta_reset()
        s0, _ = ctx.delta_end()
        ctx.update_direct(0.1, 1, None)
        ctx.delta_begin()
        s1, _ = ctx.delta_end()
        ctx.delta_reset()
        ctx.delta_begin()
        s2, _ = ctx.delta_end()
        assert (s0[5], s1[5], s2[5]) == (1, 0, 1)
        d = nb.DeltaDecoder()
        d.apply(s2)                                     # a key frame stands alone
        p, _, _, _ = ctx.download()
        assert _same_bits(d.positions(), p)
        ctx.upload(pos[:100], vel[:100], np.ones(100, np.uint32))      # other bodies: a new sequence by itself
        ctx.delta_begin()
        s3, _ = ctx.delta_end()
        assert s3[5] == 1 and s3 == dc.Encoder().encode(pos[:100], step=1)      # `updates` counts on across uploads
        ctx.upload(np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint32))
        ctx.delta_begin()
        s4, _ = ctx.delta_end()
        assert s4 == dc.Encoder().encode(np.zeros((0, 2), np.float32), step=1)


def test_full_size_round_trip_and_ratio(nb):
    """N = 1 048 576 (BASELINE configs[2]'s body count), Barnes-Hut steps in between: size-independent property
    (decode(encode(x)) == x bit for bit, every snapshot) and the reason for the format (deltas are small)."""
    n = 1 << 20
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0002, dtype=np.float32)
    world = nb.World(pos, vel, w, method="bvh")
    dec = nb.DeltaDecoder()
    try:
        cnt = nb.Counting()
        sizes = []
        for k in range(4):
            if k:
                world.update(0.1, cnt)
            world.delta_begin()
            stream, step = world.delta_end()
            dec.apply(stream)
            assert step == k and _same_bits(dec.positions(), _by_id(world))
            sizes.append(len(stream))
        raw = n * 8
        assert sizes[0] <= nb._capi.load().nbody_delta_bound(n, 0)
        assert sizes[3] < 0.6 * raw, sizes
    finally:
        world.close()


def test_headless_driver_prints_the_experiments_two_lines_and_round_trips(nb):
    """nbody_run ... delta_every: "raw:" / "comp:" as the commented code of main.rs:124-133 would print them, and the
    C++ receiver (nbody_delta_decoder_* through csrc/world.hpp) ends with the device's positions."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(nb._capi.LIB_PATH), "nbody_run")
    if not os.path.exists(exe):
        pytest.skip("nbody_run not built")
    r = subprocess.run([exe, "7", "bvh", "777", "0", "f", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n = int(re.search(r"len: (\d+)", r.stdout).group(1))
    raw = [int(v) for v in re.findall(r"raw: (\d+)", r.stdout)]
    comp = [int(v) for v in re.findall(r"comp: (\d+)", r.stdout)]
    assert raw == [8 * n] * 3 and len(comp) == 3
    assert comp[0] > 0.9 * raw[0] and comp[2] < 0.5 * raw[0]          # key frame, then deltas
    assert f"differs from the device in 0 of {n} bodies" in r.stdout


@pytest.mark.parametrize("name,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_device_key_frame_equals_the_committed_golden_stream(nb, name, dtype):
    """tests/golden/delta_nbd1.npz: the device encoder produces the committed bytes for the committed frame (a key
    frame: the first stream after an upload)."""
    import os
    C = nb._capi
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "delta_nbd1.npz"))
    frame = g[f"{name}_frame0"]
    assert frame.dtype == dtype
    with C.Context(0) as ctx:
        ctx.upload(frame, np.zeros_like(frame), np.ones(frame.shape[0], np.uint32))
        ctx.delta_begin()
        stream, step = ctx.delta_end()
    assert step == 0 and stream == g[f"{name}_stream0"].tobytes()
