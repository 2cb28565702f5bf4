"""Several GPUs behind one handle (nbody_create_multi): the reference's single `world.update` call (main.rs:120)
sharded over devices inside the library.  Needs an MI355X.

What one GPU can prove, and does here:
  * n_devices = 1 under RCCL (communicator, in-place ncclAllGather of one rank, streams and events all execute):
    bit-identical to the plain context, direct and tree steps;
  * the sharding itself — block layout, chunked exchange on the communication stream, slices of the tree-ordered
    targets, packed {row, position, velocity} exchange, lazy velocity gather, replica refresh — rehearsed with the one
    physical device listed 2, 3, 4 and 8 times under the peer-copy exchange: bit-identical to the plain context for tree
    steps and EXACT arithmetic, within the frozen tolerance for FAST.
With two or more GPUs visible the RCCL exchange between distinct devices runs too (skipped otherwise; the driver's
8-GPU node is where it executes).
"""
import numpy as np
import pytest

from tests._tol import ACC_RTOL

pytestmark = pytest.mark.gpu
F32 = np.float32


def _single(nb, pos, vel, w, **params):
    c = nb._capi.Context(0)
    if params:
        c.set_params(**params)
    c.upload(pos, vel, w)
    return c


def _multi(nb, devices, pos, vel, w, exchange=None, chunks=0, **params):
    c = nb._capi.MultiContext(devices, exchange, chunks)
    if params:
        c.set_params(**params)
    c.upload(pos, vel, w)
    return c


def _same_rows(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _n_gpus():
    import torch
    return torch.cuda.device_count()


# ------------------------------------------------------------------ one device, RCCL: identical to the plain context
@pytest.mark.parametrize("n", [1, 1000, 20000, 150000])
def test_one_device_rccl_direct_bit_identical(nb, n):
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(n, seed=301)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    with _single(nb, pos, vel, w) as s, _multi(nb, [0], pos, vel, w, C.EXCHANGE_RCCL) as m:
        assert m.multi_info()[:3] == (1, C.EXCHANGE_RCCL, 1)
        assert np.array_equal(s.accel_direct(), m.accel_direct())
        for steps in (1, 4):            # odd and even: the position buffers end up swapped or not
            s.update_direct(0.1, steps)
            m.update_direct(0.1, steps)
            assert _same_rows(s.download(), m.download())


@pytest.mark.parametrize("kind_name,dtype,order", [("bvh", F32, "as_written"), ("bvh", F32, "consistent"), ("quad", F32, "consistent"),
                                                   ("quad", np.float64, "consistent"), ("bvh", np.float64, "as_written")])
def test_one_device_rccl_tree_bit_identical(nb, lab, monkeypatch, kind_name, dtype, order):
    C = nb._capi
    monkeypatch.setenv("NBODY_MULTI_FORCE_EXCHANGE", "1")   # the sliced step + the one-rank all-gather, not the shortcut
    kind = C.TREE_BVH if kind_name == "bvh" else C.TREE_QUAD
    n = 30011
    pos, vel, _ = nb.scenes.plummer(n, seed=302, dtype=dtype)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    prm = dict(theta=0.7, order=C.ORDER_AS_WRITTEN if order == "as_written" else C.ORDER_CONSISTENT)
    with _single(nb, pos, vel, w, **prm) as s, _multi(nb, [0], pos, vel, w, C.EXCHANGE_RCCL, **prm) as m:
        cs, cm = C.Counting(), C.Counting()
        s.update_tree(kind, 0.1, 3, cs)
        m.update_tree(kind, 0.1, 3, cm)
        assert _same_rows(s.download(), m.download())
        assert cm.build_bvh > 0 and cm.sum_gravity > 0 and cm.post_calculations > 0


# ------------------------------------------------------------------ the sharding, rehearsed on one physical device
@pytest.mark.parametrize("ranks,chunks,n", [(2, 1, 4096), (2, 3, 5000), (3, 2, 10007), (4, 1, 1000), (8, 2, 70001), (4, 4, 63)])
def test_rehearsal_direct_exact_bit_identical(nb, orc, ranks, chunks, n):
    """EXACT arithmetic is one sequential chain per target: whoever computes a target gets the reference's bits, so
    the sharded trajectory must equal the plain context's (and the oracle's) exactly — ragged sizes, several chunks."""
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(n, seed=303)
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    with _single(nb, pos, vel, w, arith=C.ARITH_EXACT) as s, \
            _multi(nb, [0] * ranks, pos, vel, w, C.EXCHANGE_PEER, chunks, arith=C.ARITH_EXACT) as m:
        g, x, c, block = m.multi_info()
        assert (g, x, c) == (ranks, C.EXCHANGE_PEER, chunks) and block % 64 == 0 and g * c * block >= n
        for steps in (1, 2):
            s.update_direct(0.1, steps)
            m.update_direct(0.1, steps)
            assert _same_rows(s.download(), m.download())
    if n <= 10007:
        rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=3, nthreads=8)
        with _multi(nb, [0] * ranks, pos, vel, w, C.EXCHANGE_PEER, chunks, arith=C.ARITH_EXACT) as m:
            m.update_direct(0.1, 3)
            p, v, _, _ = m.download()
            assert np.array_equal(p, rp) and np.array_equal(v, rv)


@pytest.mark.parametrize("ranks,chunks", [(2, 1), (4, 2), (8, 1)])
def test_rehearsal_direct_fast_within_tolerance(nb, orc, ranks, chunks):
    """FAST sums in another order per shard shape: the velocity change of one step is dt * a, with a inside the frozen
    tolerance of the exactly accumulated sum (tests/_tol.py).  Bodies start at rest so that v1 = fl(a * dt) exactly."""
    C = nb._capi
    n = 50000
    pos, vel, w = nb.scenes.plummer(n, seed=304)
    vel = np.zeros_like(vel)            # from rest the new velocity IS fl(a * dt): nothing of a is absorbed by v
    ref64, norm = orc.direct_accel(pos, w, accum="f64", nthreads=16)
    with _multi(nb, [0] * ranks, pos, vel, w, C.EXCHANGE_PEER, chunks, arith=C.ARITH_FAST) as m:
        m.update_direct(0.1, 1)
        p, v, _, ids = m.download()
    assert np.array_equal(ids, np.arange(n))
    err = np.abs(v.astype(np.float64) / 0.1 - ref64).sum(axis=1)          # v1 = fl(a * 0.1f)
    slack = 4 * np.finfo(F32).eps * np.abs(ref64).sum(axis=1)
    assert np.all(err <= ACC_RTOL * norm + slack), float(((err - slack) / norm).max())
    assert np.array_equal(p, (pos + v * F32(0.1)).astype(F32))  # x += v * dt with the new velocity (main.rs:421-422)


@pytest.mark.parametrize("kind_name,dtype,order,ranks", [("bvh", F32, "as_written", 2), ("bvh", F32, "consistent", 3),
                                                         ("quad", F32, "consistent", 4), ("quad", np.float64, "consistent", 3),
                                                         ("bvh", np.float64, "consistent", 2), ("bvh", F32, "as_written", 8)])
def test_rehearsal_tree_bit_identical(nb, kind_name, dtype, order, ranks):
    C = nb._capi
    kind = C.TREE_BVH if kind_name == "bvh" else C.TREE_QUAD
    n = 20011                                                   # not a multiple of any rank count used
    pos, vel, _ = nb.scenes.plummer(n, seed=305, dtype=dtype)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    prm = dict(theta=0.6, order=C.ORDER_AS_WRITTEN if order == "as_written" else C.ORDER_CONSISTENT)
    with _single(nb, pos, vel, w, **prm) as s, _multi(nb, [0] * ranks, pos, vel, w, C.EXCHANGE_PEER, **prm) as m:
        for steps in (1, 3):
            s.update_tree(kind, 0.1, steps)
            m.update_tree(kind, 0.1, steps)
            assert _same_rows(s.download(), m.download())


def test_rehearsal_mixed_calls_keep_the_replicas_whole(nb):
    """Direct steps leave each device with the velocities of its own blocks; a tree step, a download, a snapshot, a frame
    or a delta stream needs whole rows (gathered lazily), and a parity hook that permutes the first device's rows
    (accel_tree on the BVH) is followed by a refresh of the others.  The sequence below equals the plain context's."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[:30000], vel[:30000], w[:30000]
    prm = dict(arith=C.ARITH_EXACT, theta=2.0)
    with _single(nb, pos, vel, w, **prm) as s, _multi(nb, [0, 0, 0], pos, vel, w, C.EXCHANGE_PEER, 2, **prm) as m:
        for ctx in (s, m):
            ctx.update_direct(0.1, 2)
            ctx.update_tree(C.TREE_BVH, 0.1, 2)
            ctx.update_direct(0.1, 1)
        assert _same_rows(s.download(), m.download())
        assert np.array_equal(s.render(), m.render())
        a_s, a_m = s.accel_tree(C.TREE_BVH), m.accel_tree(C.TREE_BVH)      # permutes rows
        assert np.array_equal(a_s, a_m)
        assert s.tree_info().n_nodes == m.tree_info().n_nodes
        for ctx in (s, m):
            ctx.update_tree(C.TREE_QUAD, 0.1, 1)
            ctx.update_direct(0.1, 2)
            ctx.snapshot_begin()
            ctx.update_direct(0.1, 1)                                       # overlaps the hand-off
        snap_s, snap_m = s.snapshot_end(), m.snapshot_end()
        assert _same_rows(snap_s[:4], snap_m[:4]) and snap_s[4] == snap_m[4] == 8
        assert _same_rows(s.download(), m.download())
        ds, dm = nb.DeltaDecoder(), nb.DeltaDecoder()
        for k in range(3):
            for ctx, dec in ((s, ds), (m, dm)):
                ctx.delta_begin()
                stream, _ = ctx.delta_end()
                dec.apply(stream)
                ctx.update_tree(C.TREE_BVH, 0.1, 1)
            assert np.array_equal(ds.positions(), dm.positions())
        assert m.n == s.n == 30000


def test_multi_refuses_what_it_cannot_do(nb):
    C = nb._capi
    with pytest.raises(C.NBodyError, match="one rank per physical device"):
        C.MultiContext([0, 0], C.EXCHANGE_RCCL)
    with pytest.raises(C.NBodyError):
        C.MultiContext([0, 99], C.EXCHANGE_PEER)
    with pytest.raises(C.NBodyError):
        C.MultiContext([], C.EXCHANGE_PEER)
    pos, vel, w = nb.scenes.plummer(2000, seed=306)
    with _multi(nb, [0, 0], pos, vel, w, C.EXCHANGE_PEER) as m:
        with pytest.raises(C.NBodyError, match="shards its steps itself"):
            m.update_tree_shard(C.TREE_QUAD, 0.1, 0, 1000)
        m.upload(pos.astype(np.float64), vel.astype(np.float64), w)
        with pytest.raises(C.NBodyError, match="f32"):
            m.update_direct(0.1, 1)          # the direct path is f32, as for one device
        m.upload(pos, vel, w)                # and recovers
        m.update_direct(0.1, 1)
        assert m.download()[0].shape == (2000, 2)
    with _multi(nb, [0], np.zeros((0, 2), F32), np.zeros((0, 2), F32), np.zeros(0, np.uint32), C.EXCHANGE_PEER) as m:
        m.update_direct(0.1, 2)              # empty world: nothing to do, nothing to hang on
        m.update_tree(C.TREE_BVH, 0.1, 1)
        assert m.download()[0].shape == (0, 2)


def test_config5_shard_layout_rehearsal(nb, orc):
    """BASELINE config 5's layout (8 ranks, several chunks per step) at a size one GPU steps in seconds: 8 x the one
    device, 262 144 bodies, chunks forced to 4, FAST arithmetic, sampled targets against the oracle."""
    C = nb._capi
    n = 262144
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0005)
    vel = np.zeros_like(vel)
    tg = np.arange(0, n, 64)
    ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
    with _multi(nb, [0] * 8, pos, vel, w, C.EXCHANGE_PEER, 4) as m:
        assert m.multi_info() == (8, C.EXCHANGE_PEER, 4, 8192)
        m.update_direct(0.1, 1)
        p, v, _, _ = m.download()
    err = np.abs(v[tg].astype(np.float64) / 0.1 - ref64).sum(axis=1)
    slack = 4 * np.finfo(F32).eps * np.abs(ref64).sum(axis=1)
    assert np.all(err <= ACC_RTOL * norm + slack)


def test_short_last_block_takes_a_larger_source_split(nb, orc):
    """a shard's LAST block can be shorter than the others and then wants a larger blockIdx.y source split
    (262 144 rows -> 8, 261 944 rows -> 9), whose partial sums did not fit an area sized from the full block: the step
    failed with NBODY_ERR_INVALID for N = 2^21 - 200 on 2 ranks x 4 chunks.  One whole step, sampled targets from every
    block (the short one included) against the oracle."""
    C = nb._capi
    n = (1 << 21) - 200
    pos, vel, w = nb.scenes.plummer(n, seed=309)
    vel = np.zeros_like(vel)
    with _multi(nb, [0, 0], pos, vel, w, C.EXCHANGE_PEER, 4) as m:
        g, x, c, block = m.multi_info()
        assert (g, c, block) == (2, 4, 262144) and n - 7 * block == 261944
        m.update_direct(0.1, 1)
        p, v, _, ids = m.download()
    assert np.array_equal(ids, np.arange(n, dtype=np.uint32))
    tg = np.concatenate([np.arange(77, n, 4099), np.arange(n - 2000, n, 97)])
    ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
    err = np.abs(v[tg].astype(np.float64) / 0.1 - ref64).sum(axis=1)
    slack = 4 * np.finfo(F32).eps * np.abs(ref64).sum(axis=1)
    assert np.all(err - slack <= ACC_RTOL * norm), float(((err - slack) / norm).max())
    assert np.array_equal(p, (pos + v * F32(0.1)).astype(F32))


# ------------------------------------------------------------------ distinct devices under RCCL (needs >= 2 GPUs)
def _overlap_lines(err):
    """NBODY_TRACE's lines of a sharded direct call's last step: (rank, chunk, lead_us, took_us) — the chunk's gather began
    `lead_us` before the NEXT chunk's kernels ended (csrc/multi.hip, direct_worker)."""
    import re
    out = []
    for m in re.finditer(r"\[nbody\] multi: rank (\d+) \(device \d+\) chunk (\d+): gather began (-?[0-9.]+) us before chunk \d+'s kernels "
                         r"ended, took ([0-9.]+) us", err):
        out.append((int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    return out


def test_chunk_gathers_start_while_the_next_chunk_computes(nb, monkeypatch, capfd):
    """the exchange of chunk c runs on a communication stream of its own, ordered after chunk c's kernels by an
    event, while the compute stream goes on with chunk c + 1.  Rehearsed on the one GPU of the test box (two ranks sharing the
    device, peer-copy exchange): for every rank and every chunk but the last, the gather BEGINS before the next chunk's kernels
    end — by most of a chunk's kernel time, since nothing but that chunk's own kernels stands before it.  The event timestamps
    come from the library under NBODY_TRACE (timed events recorded on both streams during the call's last step)."""
    C = nb._capi
    n, ranks, chunks = 262144, 2, 4
    pos, vel, w = nb.scenes.plummer(n, seed=311)
    monkeypatch.setenv("NBODY_TRACE", "1")
    with _multi(nb, [0] * ranks, pos, vel, w, C.EXCHANGE_PEER, chunks) as m:
        m.update_direct(0.1, 2)
        capfd.readouterr()
        timer = C.Timer()
        m.set_timer(timer)
        m.update_direct(0.1, 2)
        kms, _ = timer.read()
        m.set_timer(None)
        err = capfd.readouterr().err
    lines = _overlap_lines(err)
    assert sorted((r, c) for r, c, _, _ in lines) == [(r, c) for r in range(ranks) for c in range(chunks - 1)], err[-1500:]
    for r, c, lead_us, took_us in lines:
        assert lead_us > 0.0, (r, c, lead_us)              # the gather did not wait for the next chunk's kernels
    # the typical lead is a chunk's kernel time (HIP events around device 0's main passes); demand a quarter of it of most lines
    good = [lead for _, _, lead, _ in lines if lead > 0.25 * 1e3 * kms]
    assert len(good) >= len(lines) // 2, (lines, kms)


def test_rccl_between_distinct_devices(nb, monkeypatch, capfd):
    if _n_gpus() < 2:
        pytest.skip("one GPU visible: RCCL between distinct devices runs on the driver's multi-GPU node")
    C = nb._capi
    # the comm-stream overlap between DISTINCT devices: every chunk's all-gather begins before the next
    # chunk's kernels end, on every rank
    monkeypatch.setenv("NBODY_TRACE", "1")
    g0 = min(_n_gpus(), 4)
    posb, velb, wb = nb.scenes.plummer(524288, seed=312)
    with _multi(nb, list(range(g0)), posb, velb, wb, C.EXCHANGE_RCCL, 4) as mb:
        mb.update_direct(0.1, 1)
        capfd.readouterr()
        mb.update_direct(0.1, 2)
        lines = _overlap_lines(capfd.readouterr().err)
    assert sorted((r, c) for r, c, _, _ in lines) == [(r, c) for r in range(g0) for c in range(3)]
    assert all(lead > 0.0 for _, _, lead, _ in lines), lines
    monkeypatch.delenv("NBODY_TRACE")
    g = min(_n_gpus(), 4)
    n = 100003
    pos, vel, _ = nb.scenes.plummer(n, seed=307)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    prm = dict(arith=C.ARITH_EXACT, theta=0.8)
    with _single(nb, pos, vel, w, **prm) as s, _multi(nb, list(range(g)), pos, vel, w, C.EXCHANGE_RCCL, 2, **prm) as m, \
            _multi(nb, list(range(g)), pos, vel, w, C.EXCHANGE_PEER, 2, **prm) as mp_:
        for ctx in (s, m, mp_):
            ctx.update_direct(0.1, 2)
            ctx.update_tree(C.TREE_BVH, 0.1, 2)
            ctx.update_tree(C.TREE_QUAD, 0.1, 1)
            ctx.update_direct(0.1, 1)
        ref = s.download()
        assert _same_rows(ref, m.download()) and _same_rows(ref, mp_.download())


def test_config5_as_stated_rehearsed_on_one_gpu(nb, orc):
    """BASELINE config 5 AS STATED — 16 777 216 bodies, direct O(N^2) f32, sharded over 8 ranks, positions exchanged every
    step — with the 8 ranks on the one GPU of the test box (peer-copy exchange; the RCCL exchange needs 8 physical devices
    and runs in the driver's scaling bench).  One whole step: 2.8e14 pair interactions, 8 x 8 blocks of 262 144 targets
    (the chunk count the size picks by itself), every block's new positions copied to the 7 other replicas, then the lazy
    velocity gather of the download.  Checked on 4 096 sampled targets against the oracle (frozen tolerance), the
    integration bit for bit given the new velocities, and ids untouched."""
    C = nb._capi
    n = 1 << 24
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0005)
    vel = np.zeros_like(vel)                       # from rest: v1 = fl(a * dt), so the step exposes the accelerations
    with _multi(nb, [0] * 8, pos, vel, w, C.EXCHANGE_PEER) as m:
        assert m.multi_info() == (8, C.EXCHANGE_PEER, 8, 262144)
        cnt = C.Counting()
        m.update_direct(0.1, 1, cnt)
        p, v, w2, ids = m.download()
    print(f"[config 5 on one GPU] one step of 16.7M bodies on 8 ranks sharing the device: {cnt.sum_gravity:.1f} s")
    assert np.array_equal(ids, np.arange(n, dtype=np.uint32)) and np.array_equal(w2, w)
    assert np.array_equal(p, (pos + v * F32(0.1)).astype(F32))
    tg = np.arange(1234, n, 4096)                  # 4 096 targets spread over every rank's blocks
    ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
    err = np.abs(v[tg].astype(np.float64) / 0.1 - ref64).sum(axis=1)
    slack = 4 * np.finfo(F32).eps * np.abs(ref64).sum(axis=1)
    rel = (err - slack) / norm
    print(f"[config 5 on one GPU] e_gpu on {len(tg)} sampled targets: median {np.median(err / norm):.2e} max {(err / norm).max():.2e}")
    assert np.all(rel <= ACC_RTOL), float(rel.max())


def test_one_device_takes_the_single_device_step_driver(nb, monkeypatch, capfd):
    """With one device there is nothing to shard: tree steps on a multi context run the single-device driver (f32 BVH steps
    enqueued ahead of the host), not the sliced step + exchange."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::3].copy(), vel[::3].copy(), w[::3].copy()
    monkeypatch.setenv("NBODY_TRACE", "1")
    with _single(nb, pos, vel, w) as s, _multi(nb, [0], pos, vel, w, C.EXCHANGE_PEER) as m:
        cm = C.Counting()
        s.update_tree(C.TREE_BVH, 0.1, 5)
        capfd.readouterr()
        m.update_tree(C.TREE_BVH, 0.1, 5, cm)
        assert capfd.readouterr().err.count("step ahead: build verdict 1") == 4
        assert _same_rows(s.download(), m.download()) and cm.build_bvh > 0 and m.counting().sum_gravity > 0


def test_contexts_give_their_device_memory_back(nb):
    """Create, step (direct, BVH ahead of the host, quad, f64 BVH on the device), destroy — thirty times, single-device and
    multi-device: the device's free memory ends where it started (events, streams, pinned buffers, workers included)."""
    import torch
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(30000, seed=308)
    p64, v64 = pos.astype(np.float64), vel.astype(np.float64)

    def cycle(make):
        with make() as c:
            c.upload(pos, vel, w)
            c.update_direct(0.1, 2)
            c.update_tree(C.TREE_BVH, 0.1, 3)
            c.update_tree(C.TREE_QUAD, 0.1, 1)
            c.snapshot_begin()
            c.snapshot_end()
            c.upload(p64, v64, w)
            c.update_tree(C.TREE_BVH, 0.1, 2)

    for make in (lambda: C.Context(0), lambda: C.MultiContext([0, 0, 0], C.EXCHANGE_PEER, 2), lambda: C.MultiContext([0], C.EXCHANGE_RCCL)):
        cycle(make)                                     # first use: lazily created runtime state (RCCL, code objects) stays
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        for _ in range(30):
            cycle(make)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info(0)
        assert free0 - free1 < 64 << 20, (free0 - free1) / 2**20
