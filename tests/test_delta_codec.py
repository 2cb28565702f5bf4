"""Delta-snapshot stream "NBD1": the numpy statement of the format (oracle/delta_codec.py) against itself and against
the library's host decoder (nbody_delta_decoder_*; no device needed).  Bit-exact: it is integer work."""
import struct

import numpy as np
import pytest

from oracle import delta_codec as dc


def _walk(rng, n, dtype, steps, scale=1e5):
    pos = (rng.random((n, 2)) * scale).astype(dtype)
    vel = rng.standard_normal((n, 2)).astype(dtype)
    out = []
    for _ in range(steps):
        out.append(pos.copy())
        vel = (vel + rng.standard_normal((n, 2)) * 0.01).astype(dtype)
        pos = (pos + vel * dtype(0.1)).astype(dtype)
    return out


def _same_bits(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 700])
def test_round_trip_and_host_decoder(nb, dtype, n):
    rng = np.random.default_rng(n + 5)
    enc, dec, cdec = dc.Encoder(), dc.Decoder(), nb.DeltaDecoder()
    for k, pos in enumerate(_walk(rng, n, dtype, 5)):
        s = enc.encode(pos, step=10 * k)
        assert len(s) <= nb._capi.load().nbody_delta_bound(n, int(dtype == np.float64))
        assert s[5] == (1 if k == 0 else 0)
        dec.apply(s)
        cdec.apply(s)
        assert _same_bits(dec.positions(), pos)
        assert cdec.n == n and cdec.step == 10 * k and cdec.dtype == dtype
        assert _same_bits(cdec.positions(), pos)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_bit_pattern_survives(nb, dtype):
    """NaNs with payloads, infinities, both zeros, subnormals, sign changes, the extremes: the stream is lossless."""
    U = np.uint32 if dtype == np.float32 else np.uint64
    rng = np.random.default_rng(3)
    n = 300
    info = np.finfo(dtype)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, info.max, info.min, info.tiny, -info.tiny,
                        info.smallest_subnormal, -info.smallest_subnormal, 1.0, -1.0], dtype)
    frames = []
    for k in range(4):
        bits = rng.integers(0, np.iinfo(U).max, (n, 2), dtype=U, endpoint=True)
        pos = bits.view(dtype).copy()
        pos[:special.size, k % 2] = np.roll(special, k)
        frames.append(pos)
    enc, cdec = dc.Encoder(), nb.DeltaDecoder()
    for pos in frames:
        s = enc.encode(pos)
        cdec.apply(s)
        assert _same_bits(cdec.positions(), pos)


def test_uniform_motion_costs_nothing(nb):
    """Second-order prediction: keys that advance by the same amount every snapshot leave zero residuals."""
    n = 640
    base = (np.arange(2 * n, dtype=np.float32).reshape(n, 2) + 4096.0)       # one binade: equal steps = equal key steps
    frames = [base + np.float32(0.25 * k) for k in range(4)]
    enc, cdec = dc.Encoder(), nb.DeltaDecoder()
    sizes = []
    for pos in frames:
        s = enc.encode(pos.astype(np.float32))
        cdec.apply(s)
        sizes.append(len(s))
        assert _same_bits(cdec.positions(), pos.astype(np.float32))
    nblk = n // 64
    wb = (2 * nblk + 7) // 8 * 8
    assert sizes[2] == sizes[3] == dc.HEADER + wb                 # widths only, every width 0 with predictor 1
    assert sizes[1] > sizes[2] and sizes[0] > sizes[1]


def test_reset_starts_a_new_sequence(nb):
    rng = np.random.default_rng(9)
    frames = _walk(rng, 200, np.float32, 4)
    enc = dc.Encoder()
    s0 = enc.encode(frames[0])
    s1 = enc.encode(frames[1])
    enc.reset()
    s2 = enc.encode(frames[2])
    assert (s0[5], s1[5], s2[5]) == (1, 0, 1)
    late = nb.DeltaDecoder()          # a receiver that joins at the second key frame
    late.apply(s2)
    assert _same_bits(late.positions(), frames[2])


def test_host_decoder_rejects_bad_streams_and_keeps_its_state(nb):
    C = nb._capi
    rng = np.random.default_rng(11)
    frames = _walk(rng, 130, np.float32, 3)
    enc = dc.Encoder()
    s0, s1, s2 = (enc.encode(f, step=i) for i, f in enumerate(frames))
    d = nb.DeltaDecoder()
    with pytest.raises(C.NBodyError, match="before any key frame"):
        d.apply(s1)
    with pytest.raises(C.NBodyError):
        d.positions()
    d.apply(s0)
    wb = (2 * 3 + 7) // 8 * 8
    bad = [
        (b"", "shorter"),
        (s1[:31], "shorter"),
        (b"XBD1" + s1[4:], "magic"),
        (s1[:4] + bytes([16]) + s1[5:], "header"),
        (s1[:6] + b"\x01" + s1[7:], "header"),
        (s1[:-8], "size"),
        (s1 + bytes(8), "size"),
        (s1[:8] + struct.pack("<Q", 131) + s1[16:], "another body count"),
        (s1[:dc.HEADER] + bytes([s1[dc.HEADER] + 1]) + s1[dc.HEADER + 1:], "add up"),
        (s1[:dc.HEADER] + bytes([33]) + s1[dc.HEADER + 1:], "width exceeds"),
        (s1[:dc.HEADER + wb - 1] + b"\x01" + s1[dc.HEADER + wb:], "padding"),
    ]
    for stream, why in bad:
        with pytest.raises(C.NBodyError, match=why):
            d.apply(stream)
        assert d.step == 0 and _same_bits(d.positions(), frames[0])     # untouched
    f64 = dc.Encoder().encode(frames[0].astype(np.float64))
    d.apply(s1)
    with pytest.raises(C.NBodyError, match="precision"):
        d.apply(dc.Encoder().encode(frames[0].astype(np.float64))[:5] + b"\x00" + f64[6:])
    d.apply(s2)
    assert d.step == 2 and _same_bits(d.positions(), frames[2])
    with pytest.raises(C.NBodyError):
        C.DeltaDecoder().positions()


def test_host_decoder_caps_the_body_count_a_header_may_claim(nb):
    """An untrusted key-frame header may claim 2^31 - 1 bodies with all-zero widths: a valid 67 MB stream that would
    need ~100 GB of decoder state.  With a caller-set cap it is refused before anything is allocated."""
    C = nb._capi
    rng = np.random.default_rng(12)
    frames = _walk(rng, 100, np.float32, 1)
    d = nb.DeltaDecoder()
    d.apply(dc.Encoder().encode(frames[0], step=7))
    d.set_max_bodies(1 << 20)
    huge = (1 << 31) - 1
    nblk = (huge + 63) // 64
    wb = (2 * nblk + 7) // 8 * 8
    forged = b"NBD1" + bytes([32, 1, 0, 0]) + struct.pack("<QQQ", huge, 0, 0) + bytes(wb)
    with pytest.raises(C.NBodyError, match="out of range"):
        d.apply(forged)
    assert d.step == 7 and d.n == 100 and _same_bits(d.positions(), frames[0])
    d.set_max_bodies(99)                                   # the cap also binds honest streams
    with pytest.raises(C.NBodyError, match="out of range"):
        d.apply(dc.Encoder().encode(frames[0], step=8))
    assert d.step == 7


def test_bound_is_the_worst_case(nb):
    lib = nb._capi.load()
    for n in (0, 1, 64, 65, 1000):
        for f64 in (0, 1):
            nblk = (n + 63) // 64
            assert lib.nbody_delta_bound(n, f64) == 32 + (2 * nblk + 7) // 8 * 8 + 2 * nblk * (64 if f64 else 32) * 8
    assert lib.nbody_delta_bound(-1, 0) == 0
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2**32 - 1, (128, 2), dtype=np.uint32)          # random bits: every block needs full width
    e = dc.Encoder()
    e.encode(np.zeros((128, 2), np.float32))
    assert len(e.encode(bits.view(np.float32))) <= lib.nbody_delta_bound(128, 0)


def test_golden_streams_pin_the_format(nb):
    """tests/golden/delta_nbd1.npz (make_golden_delta.py): the statement of the format still encodes the committed
    frames to the committed bytes, and the library's decoder reads those bytes back to the frames."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "delta_nbd1.npz"))
    for name in ("f32", "f64"):
        enc, cdec, dec = dc.Encoder(), nb.DeltaDecoder(), dc.Decoder()
        for k in range(int(g[f"{name}_steps"])):
            frame, stream = g[f"{name}_frame{k}"], g[f"{name}_stream{k}"].tobytes()
            assert enc.encode(frame, step=7 * k) == stream, (name, k)
            cdec.apply(stream)
            dec.apply(stream)
            assert cdec.step == 7 * k
            assert _same_bits(cdec.positions(), frame) and _same_bits(dec.positions(), frame), (name, k)
