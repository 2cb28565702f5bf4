"""Parity of the direct O(N^2) HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest

from tests._tol import TRAJ_ATOL_POS, check_fast

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def ctx(nb):
    c = nb._capi.Context(0)
    yield c
    c.close()


def _refs(orc, pos, w, targets=None, nthreads=16):
    ref64, norm = orc.direct_accel(pos, w, targets=targets, accum="f64", nthreads=nthreads)
    cpu32, _ = orc.direct_accel(pos, w, targets=targets, nthreads=nthreads)
    return ref64, norm, cpu32


# ------------------------------------------------------------------ EXACT arithmetic: bit-identical
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257, 1000, 4096])
def test_exact_accel_bit_identical_to_oracle(nb, orc, ctx, n):
    pos, vel, _ = nb.scenes.plummer(n, seed=41)
    w = (np.arange(n) % 11 + 1).astype(np.uint32)
    ctx.set_params(arith=nb._capi.ARITH_EXACT)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    ref, _ = orc.direct_accel(pos, w, nthreads=8)
    assert np.array_equal(acc, ref.astype(F32))


def test_exact_update_bit_identical_over_steps(nb, orc, ctx):
    pos, vel, w = nb.scenes.plummer(1024, seed=42)
    ctx.set_params(arith=nb._capi.ARITH_EXACT)
    ctx.upload(pos, vel, w)
    ctx.update_direct(0.1, 10)
    p, v, w2, ids = ctx.download()
    rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=10, nthreads=8)
    assert np.array_equal(p, rp) and np.array_equal(v, rv)
    assert np.array_equal(ids, np.arange(1024)) and np.array_equal(w2, w)


def test_exact_skip_semantics_edge_cases(nb, orc, ctx):
    """main.rs:241-243: zero, subnormal, inf and NaN sums leave the accumulator untouched; the smallest normal sum
    does not.  Checked on the device in EXACT arithmetic, bit for bit."""
    tiny = np.finfo(F32).tiny
    pos = np.array([[0, 0], [0, 0],                    # coincident pair
                    [1e-39, 0], [0, -1e-40],           # subnormal separations from the origin
                    [tiny, 0],                         # smallest normal separation
                    [np.inf, 1], [np.nan, 2],          # non-finite
                    [3e38, 3e38], [-3e38, -3e38],      # finite, but |dx|+|dy| overflows
                    [3, 4], [0.01, 0.0], [100, -50]], F32)
    vel = np.zeros_like(pos)
    w = np.array([2, 3, 1, 1, 1, 5, 5, 7, 7, 2, 1, 750000], np.uint32)
    ctx.set_params(arith=nb._capi.ARITH_EXACT)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    ref, _ = orc.direct_accel(pos, w)
    assert np.array_equal(acc, ref.astype(F32), equal_nan=True)
    assert np.all(np.isfinite(acc[[0, 1, 2, 3, 4, 9, 10, 11]]))


def test_auto_switches_to_exact_on_hazardous_positions(nb, orc, ctx):
    """AUTO = FAST unless a coordinate is non-finite, >= 2^60, or non-zero below 2^-22; then EXACT (bit-exact)."""
    pos, vel, w = nb.scenes.plummer(2048, seed=43)
    for poison in (np.nan, np.inf, 2.0 ** 61, 1e-30, -1e-12):
        p = pos.copy()
        p[777, 1] = poison
        ctx.set_params(arith=nb._capi.ARITH_AUTO)
        ctx.upload(p, vel, w)
        acc = ctx.accel_direct()
        ref, _ = orc.direct_accel(p, w, nthreads=8)
        assert np.array_equal(acc, ref.astype(F32), equal_nan=True), poison
    # zero coordinates are not hazardous (differences are then 0 or >= 2^-46)
    p = pos.copy()
    p[5] = (0.0, 0.0)
    p[6] = (0.0, 123.0)
    ctx.upload(p, vel, w)
    acc = ctx.accel_direct()
    ref64, norm, cpu32 = _refs(orc, p, w)
    check_fast(acc, ref64, norm, cpu32)
    assert not np.array_equal(acc, cpu32.astype(F32))  # i.e. FAST really ran


def test_tiny_clamp_forces_exact(nb, orc, ctx):
    pos, vel, w = nb.scenes.plummer(512, seed=44)
    ctx.set_params(arith=nb._capi.ARITH_FAST, clamp=1e-9)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    ref, _ = orc.direct_accel(pos, w, clamp=1e-9)
    assert np.array_equal(acc, ref.astype(F32))
    ctx.set_params(clamp=0.001)


# ------------------------------------------------------------------ FAST arithmetic: stated tolerance
@pytest.mark.parametrize("n", [1, 7, 64, 100, 1000, 4097, 20000])
def test_fast_accel_within_tolerance(nb, orc, ctx, n):
    pos, vel, _ = nb.scenes.plummer(n, seed=45)
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    ctx.set_params(arith=nb._capi.ARITH_FAST)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    check_fast(acc, *_refs(orc, pos, w))


def test_fast_coincident_and_clamped_pairs(nb, orc, ctx):
    """Coincident bodies contribute exactly nothing (the 2^-90 bias keeps 0 * rcp finite) and near pairs hit the
    0.001 clamp, in FAST arithmetic."""
    pos, vel, w = nb.scenes.plummer(3000, seed=46)
    pos[100] = pos[200]                       # exact duplicate
    pos[300] = pos[400] + F32(0.0078125)      # d^2 = 6.1e-5 < 0.001
    w = (np.arange(3000) % 3 + 1).astype(np.uint32)
    ctx.set_params(arith=nb._capi.ARITH_FAST)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    assert np.all(np.isfinite(acc))
    check_fast(acc, *_refs(orc, pos, w))


def test_fast_mixed_masses_reference_scene(nb, orc, ctx):
    """Masses 1 / 750 000 / 75 000 000 as in World::new (main.rs:282-291)."""
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[:20000], vel[:20000], w[:20000]
    ctx.set_params(arith=nb._capi.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    check_fast(acc, *_refs(orc, pos, w))


def test_config2_65536_all_targets(nb, orc, ctx):
    """BASELINE config 2: 65 536 bodies direct f32, every target compared with the CPU oracle."""
    n = 65536
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0002)
    ctx.set_params(arith=nb._capi.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    rg, rc = check_fast(acc, *_refs(orc, pos, w))
    # the f32 sequential reference order is itself further from the exactly accumulated sum than the GPU is
    print(f"max err/norm: gpu {rg:.3e}, cpu f32 sequential {rc:.3e}")
    assert rg <= rc


def test_config3_1M_sampled_targets_and_properties(nb, orc, ctx):
    """BASELINE config 3: 1 048 576 bodies; 4 096 sampled targets against the oracle, plus size-independent
    properties of one full step: determinism, and total momentum change = sum of m*a*dt (equal masses here)."""
    n = 1 << 20
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0003)
    ctx.set_params(arith=nb._capi.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    tg = np.arange(0, n, 256)
    rg, rc = check_fast(acc[tg], *_refs(orc, pos, w, targets=tg))
    print(f"1M sampled: max err/norm gpu {rg:.3e}, cpu f32 sequential {rc:.3e}")
    acc2 = ctx.accel_direct()
    assert np.array_equal(acc, acc2)                      # bitwise deterministic
    # Newton's third law holds for this force law too (term_ij = -term_ji up to rounding): net force ~ 0
    net = np.abs(acc.astype(np.float64).sum(axis=0))
    assert np.all(net <= 1e-4 * np.abs(acc.astype(np.float64)).sum(axis=0))
    ctx.update_direct(0.1, 1)
    p1, v1, _, _ = ctx.download()
    dt = F32(0.1)
    v_exp = vel + acc * dt
    assert np.array_equal(v1, v_exp)                      # integrate is the reference's mul-then-add, bit exact
    assert np.array_equal(p1, pos + v_exp * dt)


def test_free_masses_1M_streamed_sampled_and_properties(nb, orc, ctx):
    """free per-body masses (main.rs:193-198, `weight as f32` at :360) at the headline's size through the
    streamed main pass (direct_stream_m: inverse masses beside the couples).  bench.py's `free_masses` leg: the same bodies and
    weights.  4 096 sampled targets against the oracle; determinism; one full step integrates as the reference writes it;
    a different kernel from the equal-mass one really ran (the accelerations differ from the masses-1 ones)."""
    n = 1 << 20
    pos, vel, _ = nb.scenes.plummer(n, seed=0x5EED0003)
    w = nb.scenes.free_weights(n, seed=0x5EED0003)
    assert len(np.unique(w)) > 1000
    ctx.set_params(arith=nb._capi.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_direct()
    tg = np.arange(0, n, 256)
    rg, rc = check_fast(acc[tg], *_refs(orc, pos, w, targets=tg), label=" free masses 1M")
    print(f"1M free masses sampled: max err/norm gpu {rg:.3e}, cpu f32 sequential {rc:.3e}")
    assert np.array_equal(acc, ctx.accel_direct())        # bitwise deterministic
    ctx.update_direct(0.1, 1)
    p1, v1, w1, _ = ctx.download()
    dt = F32(0.1)
    v_exp = vel + acc * dt
    assert np.array_equal(v1, v_exp) and np.array_equal(p1, pos + v_exp * dt) and np.array_equal(w1, w)


def test_fast_trajectory_1024x100(nb, orc, ctx):
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    ctx.set_params(arith=nb._capi.ARITH_FAST)
    ctx.upload(pos, vel, w)
    ctx.update_direct(0.1, 100)
    p, v, _, _ = ctx.download()
    rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=100, nthreads=8)
    assert np.abs(p.astype(np.float64) - rp).max() <= TRAJ_ATOL_POS


def test_direct_counter_and_timer(nb, ctx):
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(8192, seed=47)
    ctx.set_params(arith=C.ARITH_FAST)
    ctx.upload(pos, vel, w)
    t = C.Timer()
    ctx.set_timer(t)
    cnt = C.Counting()
    ctx.update_direct(0.1, 3, cnt)
    ms, launches = t.read()
    ctx.set_timer(None)
    assert launches == 3 and ms > 0
    assert cnt.sum_gravity > 0 and cnt.build_bvh == 0


def test_device_level_sharded_targets_equal_whole(nb, ctx):
    """nbody_direct_step_dev on two target shards == one call over all targets (the multi-GPU decomposition)."""
    import torch
    C = nb._capi
    n = 6000
    pos, vel, w = nb.scenes.plummer(n, seed=48)
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos).to(dev)
    tm = torch.from_numpy(w.astype(np.float32)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(begin, cnt):
        ws_bytes = C.direct_workspace_bytes(n, cnt)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        v = torch.from_numpy(vel[begin:begin + cnt].copy()).to(dev)
        out = torch.empty((cnt, 2), dtype=torch.float32, device=dev)
        acc = torch.empty((cnt, 2), dtype=torch.float32, device=dev)
        C.direct_step_dev(stream, n, tp.data_ptr(), tm.data_ptr(), begin, cnt, v.data_ptr(), out.data_ptr(),
                          acc.data_ptr(), 0.1, 0.001, C.ARITH_EXACT, ws.data_ptr(), ws_bytes)
        torch.cuda.synchronize()
        return out.cpu().numpy(), v.cpu().numpy(), acc.cpu().numpy()

    whole = run(0, n)
    a = run(0, 2500)
    b = run(2500, n - 2500)
    for k in range(3):
        assert np.array_equal(np.concatenate([a[k], b[k]]), whole[k])


# ------------------------------------------------------------------ near/far split of the sources (nearfar.hip)
def _dev_accel(nb, pos, w, arith=None, uniform=0.0):
    """Acceleration of every body through nbody_direct_step_dev; returns (acc, (hazard, fallback, n_near, state))."""
    import torch
    C = nb._capi
    n = pos.shape[0]
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(np.ascontiguousarray(pos, F32)).to(dev)
    tm = torch.from_numpy(w.astype(F32)).to(dev)
    acc = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ws_bytes = C.direct_workspace_bytes(n, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    C.direct_step_dev(stream, n, tp.data_ptr(), tm.data_ptr(), 0, n, None, None, acc.data_ptr(), 0.0, 0.001,
                      C.ARITH_AUTO if arith is None else arith, ws.data_ptr(), ws_bytes, uniform_mass=uniform)
    torch.cuda.synchronize()
    return acc.cpu().numpy(), C.direct_workspace_peek(stream, ws.data_ptr())


@pytest.fixture
def force_nearfar(monkeypatch):
    """The split is chosen by problem size (>= 2^33 pairs); these tests force it on small inputs."""
    monkeypatch.setenv("NBODY_DIRECT_NEARFAR", "2")


def test_nearfar_is_chosen_by_size(nb):
    n = 4096
    pos, vel, w = nb.scenes.plummer(n, seed=80)
    _, (_, _, _, state) = _dev_accel(nb, pos, w)
    assert state == 1                       # small problem: the single clamped pass (the split would cost more than it saves)


def test_nearfar_split_with_close_pairs_and_clusters(nb, orc, force_nearfar):
    """Bodies closer than sqrt(clamp) = 0.0316 (pairs, a tight cluster, exact duplicates) must be found 'near' and
    summed with the clamp; everything else goes through the clamp-free main pass.  Same tolerance as ever."""
    n = 30000
    pos, vel, w = nb.scenes.plummer(n, seed=81)
    rng = np.random.default_rng(3)
    pos[1000] = pos[2000] + F32(0.01)                      # a close pair
    pos[3000] = pos[4000]                                  # exact duplicate
    c = pos[5000].copy()
    for k in range(40):                                    # a cluster of 40 within 0.02
        pos[6000 + k] = c + (rng.random(2).astype(F32) - F32(0.5)) * F32(0.02)
    pos[7000] = pos[8000] + np.array([0.031, 0.0], F32)    # just inside the clamp radius
    pos[9000] = pos[9500] + np.array([0.033, 0.0], F32)    # just outside it (may or may not share a cell)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    acc, (hazard, fallback, n_near, state) = _dev_accel(nb, pos, w)
    assert (hazard, fallback, state) == (0, 0, 0)
    assert n_near >= 2 + 2 + 40 + 2                        # at least the planted ones
    check_fast(acc, *_refs(orc, pos, w))
    # every planted body must be in the near set: compare with the single clamped pass bit for bit is not
    # required (summation order differs), but the clamped pairs dominate their targets' sums:
    ref64, norm, _ = _refs(orc, pos, w, targets=np.array([1000, 2000, 3000, 4000, 6000, 6039, 7000, 8000]))
    assert np.all(np.abs(acc[[1000, 2000, 3000, 4000, 6000, 6039, 7000, 8000]] - ref64).sum(axis=1) <= 2e-5 * norm)


def test_nearfar_uniform_mass_path(nb, orc, force_nearfar):
    n = 20000
    pos, vel, w = nb.scenes.plummer(n, seed=82)
    pos[10] = pos[11] + F32(0.005)
    acc, (_, _, n_near, state) = _dev_accel(nb, pos, w, uniform=1.0)
    assert state == 0 and n_near >= 2
    check_fast(acc, *_refs(orc, pos, w))


def test_nearfar_falls_back_when_dense_or_out_of_range(nb, orc, force_nearfar):
    rng = np.random.default_rng(5)
    n = 4096
    dense = (F32(50000.0) + rng.random((n, 2)).astype(F32) * F32(0.5)).astype(F32)   # everything within 0.5 x 0.5
    w = np.ones(n, np.uint32)
    acc, (_, _, n_near, state) = _dev_accel(nb, dense, w)
    assert state == 1 and n_near > n // 64
    check_fast(acc, *_refs(orc, dense, w))
    pos, _, _ = nb.scenes.plummer(n, seed=83)
    pos[17] = (3.0e8, 1.0)                                  # beyond the cell grid's range (|x|/h >= 2^30)
    acc, (_, fallback, _, state) = _dev_accel(nb, pos, w)
    assert fallback == 1 and state == 1
    check_fast(acc, *_refs(orc, pos, w))


def test_nearfar_off_gives_the_same_answer_within_tolerance(nb, orc, monkeypatch):
    n = 16384
    pos, vel, w = nb.scenes.plummer(n, seed=84)
    pos[100] = pos[200] + F32(0.004)
    monkeypatch.setenv("NBODY_DIRECT_NEARFAR", "2")
    a_on, st_on = _dev_accel(nb, pos, w, uniform=1.0)
    monkeypatch.setenv("NBODY_DIRECT_NEARFAR", "0")
    a_off, st_off = _dev_accel(nb, pos, w, uniform=1.0)
    assert st_on[3] == 0 and st_off[3] == 1
    ref64, norm, cpu32 = _refs(orc, pos, w)
    check_fast(a_on, ref64, norm, cpu32)
    check_fast(a_off, ref64, norm, cpu32)
    # far sources see identical arithmetic in both modes; only the position of the near terms in the sum differs
    assert np.abs(a_on - a_off).sum(axis=1).max() <= 2e-5 * norm.max()


def test_nearfar_hazard_still_routes_to_exact(nb, orc, force_nearfar):
    n = 2048
    pos, vel, w = nb.scenes.plummer(n, seed=85)
    pos[5, 0] = np.nan
    acc, (hazard, _, _, state) = _dev_accel(nb, pos, w)
    assert hazard == 1 and state == 2
    ref, _ = orc.direct_accel(pos, w, nthreads=8)
    assert np.array_equal(acc, ref.astype(F32), equal_nan=True)


def test_config5_shard_slice_of_16M(nb, orc):
    """BASELINE config 5 (16 777 216 bodies sharded over 8 GPUs): on the one GPU of the test box, a slice of rank 5's
    target shard against all 16.7 M sources through the device-pointer API, checked against the oracle on sampled
    targets.  Exercises the 16.7 M-source indexing, the near/far split at that size and a mid-array target_begin."""
    import torch
    C = nb._capi
    n = 1 << 24
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0005)
    dev = torch.device("cuda:0")
    tp = torch.from_numpy(pos).to(dev)
    tm = torch.ones(n, dtype=torch.float32, device=dev)
    begin, cnt = 5 * (n // 8) + 12345, 8192
    tv = torch.from_numpy(vel[begin:begin + cnt].copy()).to(dev)
    out = torch.empty((cnt, 2), dtype=torch.float32, device=dev)
    acc = torch.empty((cnt, 2), dtype=torch.float32, device=dev)
    ws_bytes = C.direct_workspace_bytes(n, cnt)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    C.direct_step_dev(stream, n, tp.data_ptr(), tm.data_ptr(), begin, cnt, tv.data_ptr(), out.data_ptr(), acc.data_ptr(),
                      0.1, 0.001, C.ARITH_AUTO, ws.data_ptr(), ws_bytes, uniform_mass=1.0)
    torch.cuda.synchronize()
    hazard, fallback, n_near, state = C.direct_workspace_peek(stream, ws.data_ptr())
    assert (hazard, state) == (0, 0) and 0 < n_near < n // 64      # the dense core has neighbours within sqrt(clamp)
    a = acc.cpu().numpy()
    tg = np.arange(begin, begin + cnt, 128)
    rg, rc = check_fast(a[::128], *_refs(orc, pos, w, targets=tg), label=" 16M")
    # integrate: the reference's multiply-then-add, bit exact given the accelerations
    dt = F32(0.1)
    v_exp = vel[begin:begin + cnt] + a * dt
    assert np.array_equal(tv.cpu().numpy(), v_exp)
    assert np.array_equal(out.cpu().numpy(), pos[begin:begin + cnt] + v_exp * dt)


# ------------------------------------------------------------------ a few odd masses among equal ones (the reference's scene)
def test_mass_hint():
    from nbody_simulation_amd._capi import mass_hint
    assert mass_hint(np.ones(1000, np.uint32)) == 1.0
    assert mass_hint(np.full(10, 7, np.uint32)) == 7.0
    w = np.ones(151409, np.uint32)
    w[0], w[1] = 75_000_000, 750_000
    assert mass_hint(w) == -1.0                                  # World::new: two heavy bodies first (main.rs:282-291)
    assert mass_hint((np.arange(1000) % 3 + 1).astype(np.uint32)) == 0.0
    w = np.ones(1024, np.uint32); w[:5] = 9
    assert mass_hint(w) == 0.0                                   # more than n/256 odd bodies
    assert mass_hint(np.zeros(0, np.uint32)) == 0.0


def test_sparse_heavy_masses_ride_with_the_near_list(nb, orc, force_nearfar):
    """uniform_mass < 0: every mass is |uniform_mass| but for a few bodies; those are found on the device, leave the
    equal-mass main pass (their slot holds a far-away point) and are added by direct_finish with their own masses."""
    n = 40000
    pos, vel, _ = nb.scenes.plummer(n, seed=86)
    w = np.full(n, 3, np.uint32)
    heavy = [0, 1, 777, 20000, n - 1]
    w[heavy] = [75_000_000, 750_000, 1, 4_000_000_000, 2]        # heavier, lighter, beyond 2^24 (rounded by `as f32`)
    pos[100] = pos[200] + F32(0.004)                             # a genuinely near pair besides
    pos[777] = pos[778] + F32(0.002)                             # an odd-mass body that is ALSO near
    acc, (hazard, fallback, n_near, state) = _dev_accel(nb, pos, w, uniform=-3.0)
    assert (hazard, fallback, state) == (0, 0, 0)
    assert n_near == len(heavy) + 2 + 1                          # the odd masses, the pair, 777's partner
    ref64, norm, cpu32 = _refs(orc, pos, w)
    check_fast(acc, ref64, norm, cpu32)
    a0, st0 = _dev_accel(nb, pos, w, uniform=0.0)                # per-body masses: the same sum within the tolerance
    assert st0[2] == 4                                           # just the two close pairs
    check_fast(a0, ref64, norm, cpu32)


def test_reference_scene_takes_the_sparse_path(nb, orc, lab, monkeypatch):
    """The whole scene of World::new (151 k bodies, two heavy): the context finds the sparse structure on upload; the
    step equals the per-body-mass one within the tolerance, on sampled targets against the oracle."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    n = pos.shape[0]
    tg = np.concatenate([[0, 1], np.arange(2, n, 37)])
    ref64, norm, _ = _refs(orc, pos, w, targets=tg)
    accs = {}
    for no_sparse in ("0", "1"):
        monkeypatch.setenv("NBODY_DIRECT_NO_SPARSE", no_sparse)
        with C.Context(0) as c:
            c.upload(pos, vel, w)
            accs[no_sparse] = c.accel_direct()
    check_fast(accs["0"][tg], ref64, norm, label=" galaxy sparse")
    check_fast(accs["1"][tg], ref64, norm, label=" galaxy per-body")
    assert not np.array_equal(accs["0"], accs["1"])              # i.e. another kernel really ran


# ------------------------------------------------------------------ mass classes (round 3)
@pytest.mark.parametrize("n_classes", [2, 5, 32, 33])
def test_mass_classes_within_tolerance(nb, orc, lab_ctx, monkeypatch, capfd, n_classes):
    """Masses that differ freely but take few values: the sources are ordered by mass class (every class padded to whole
    1 024-source tiles) and the equal-mass kernel runs tile by tile, the class's mass in the FMA that closes a tile.  1 .. 32
    classes take that path (NBODY_TRACE says so), 33 fall back to the per-body-mass kernel; both inside the frozen tolerance,
    near bodies (clamped pairs, a coincident pair) included, and bitwise reproducible from one upload to the next."""
    C = nb._capi
    ctx = lab_ctx                                            # laboratory library: the test switches kernel variants
    n = 98304 + 777                                          # >= 65 536: the near/far split (and with it the classes) is on
    pos, vel, _ = nb.scenes.plummer(n, seed=47)
    rng = np.random.default_rng(n_classes)
    masses = np.unique(np.concatenate([[1, 7], rng.integers(1, 1 << 20, 64)]))[:n_classes].astype(np.uint32)
    assert len(masses) == n_classes
    w = masses[rng.integers(0, n_classes, n)]
    w[:n_classes] = masses                                   # every class occurs
    pos[100] = pos[200]                                      # coincident: contributes nothing
    pos[300] = pos[400] + F32(0.0078125)                     # inside the clamp radius: both are near bodies
    tg = np.arange(0, n, 37)
    ref64, norm, cpu32 = _refs(orc, pos, w, targets=tg)
    monkeypatch.setenv("NBODY_TRACE", "1")
    ctx.set_params(arith=C.ARITH_AUTO, clamp=0.001)
    ctx.upload(pos, vel, w)
    capfd.readouterr()
    a1 = ctx.accel_direct()
    err = capfd.readouterr().err
    assert (f"{n_classes} mass classes" in err) == (n_classes <= 32), err[-400:]
    check_fast(a1[tg], ref64, norm, cpu32, label=f" {n_classes} classes")
    ctx.upload(pos, vel, w)
    assert np.array_equal(ctx.accel_direct(), a1)            # reproducible
    monkeypatch.setenv("NBODY_DIRECT_NO_CLASSES", "1")       # the per-body kernel on the same input
    ctx.upload(pos, vel, w)
    a0 = ctx.accel_direct()
    check_fast(a0[tg], ref64, norm, cpu32, label=" per-body kernel")
    assert (n_classes > 32) == np.array_equal(a0, a1)


def test_mass_classes_follow_the_rows_through_a_tree_build(nb, orc, lab_ctx, monkeypatch):
    """A BVH step permutes the rows (BVHTree::from partitions in place): the class order is rebuilt for the new row order
    before the next direct step, and whole direct steps follow the per-body kernel's trajectory to rounding."""
    C = nb._capi
    ctx = lab_ctx                                            # laboratory library: the test switches kernel variants
    n = 70000
    pos, vel, _ = nb.scenes.plummer(n, seed=48)
    w = (np.arange(n) % 4 * 3 + 1).astype(np.uint32)
    ctx.set_params(arith=C.ARITH_AUTO, theta=50.0, order=C.ORDER_CONSISTENT)
    ctx.upload(pos, vel, w)
    ctx.update_direct(0.1, 1)
    ctx.update_tree(C.TREE_BVH, 0.1, 1)                      # rows permuted
    p, v, w2, ids = ctx.download()
    assert not np.array_equal(ids, np.arange(n))
    acc = ctx.accel_direct()                                 # classes of the permuted rows
    tg = np.arange(0, n, 29)
    check_fast(acc[tg], *_refs(orc, p, w2, targets=tg)[:2], label=" after a build")
    ctx.update_direct(0.1, 3)
    pa = ctx.download()
    monkeypatch.setenv("NBODY_DIRECT_NO_CLASSES", "1")
    ctx.upload(p, v, w2)
    ctx.update_direct(0.1, 3)
    pb = ctx.download()
    assert np.abs(pa[0].astype(np.float64) - pb[0]).max() <= 1e-3 and np.abs(pa[1].astype(np.float64) - pb[1]).max() <= 1e-3


def test_captured_direct_graph_is_rebuilt_when_a_tree_step_permutes_the_rows(nb, orc, ctx):
    """ADVICE r03 (capi.hip DirectGraph): four direct steps at 65 536 <= n <= 131 072 capture the step as a hipGraph with the mass
    classes' device arrays baked in.  TWO tree steps permute the rows and put every buffer of the graph's key back where it was
    while ensure_mass_classes frees and rebuilds rank / pad_slots / tile_mass: the next four direct steps must not replay the old
    capture.  Checked against the same call sequence on a fresh context that never captured before the build (eager first), and
    against the oracle's accelerations for the final rows."""
    C = nb._capi
    n = 70000
    pos, vel, _ = nb.scenes.plummer(n, seed=148)
    w = (np.arange(n) % 4 * 3 + 1).astype(np.uint32)

    def run(c, first_direct_steps):
        c.set_params(arith=C.ARITH_AUTO, theta=50.0, order=C.ORDER_CONSISTENT)
        c.upload(pos, vel, w)
        c.update_direct(0.1, first_direct_steps)              # >= 4: captured and replayed
        c.update_tree(C.TREE_BVH, 0.1, 2)                     # rows permuted twice, buffers back in place
        mid = c.download()
        c.update_direct(0.1, 4)                               # replay candidates again
        return mid, c.download()

    mid_a, end_a = run(ctx, 4)
    assert not np.array_equal(mid_a[3], np.arange(n))        # the builds did permute the rows
    with C.Context(0) as fresh:                               # same physics, but the first direct call is too short to capture:
        fresh.set_params(arith=C.ARITH_AUTO, theta=50.0, order=C.ORDER_CONSISTENT)
        fresh.upload(pos, vel, w)
        for _ in range(4):
            fresh.update_direct(0.1, 1)                       # eager steps (n_steps < 4 never captures)
        fresh.update_tree(C.TREE_BVH, 0.1, 2)
        mid_b = fresh.download()
        fresh.update_direct(0.1, 4)                           # the first capture this context makes: after the permutation
        end_b = fresh.download()
    for a, b in zip(mid_a, mid_b):
        assert np.array_equal(a, b)                           # graph replay and eager steps give the same bits
    for a, b in zip(end_a, end_b):
        assert np.array_equal(a, b)                           # ... also after the rows moved under the first graph
    # and the final state is right in absolute terms: accelerations of the final rows against the oracle
    p, v, w2, ids = end_a
    acc = ctx.accel_direct()
    tg = np.arange(0, n, 31)
    check_fast(acc[tg], *_refs(orc, p, w2, targets=tg)[:2], label=" after graph, builds, graph")


# ------------------------------------------------------------------ the main pass's variants (NBODY_DIRECT_ASM)
@pytest.mark.parametrize("n", [65536 + 16 * 5 + 3, 131072, 200003])
def test_packed_and_streamed_main_pass_agree(nb, orc, lab_ctx, monkeypatch, n):
    """NBODY_DIRECT_ASM: 0 the compiler's schedule, 1 the hand-ordered block (a pair per instruction), 2 packed couples (two
    pairs per packed op) through LDS, 3 (default) the same with the far sources streamed through SGPRs.  All four within the
    tolerance of the oracle; 2 and 3 share the arithmetic and the order of additions, hence the bits — with equal masses,
    with mass classes, with n not a multiple of anything (the far copy's padding), coincident and near bodies included."""
    C = nb._capi
    ctx = lab_ctx                                            # laboratory library: the test switches kernel variants
    pos, vel, w1 = nb.scenes.plummer(n, seed=51)
    pos[100] = pos[200]                                      # coincident: contributes nothing
    pos[300] = pos[400] + F32(0.0078125)                     # inside the clamp radius: near bodies
    tg = np.arange(0, n, 41)
    wfree = np.random.default_rng(n).integers(1, 1 << 20, n).astype(np.uint32)   # free masses: > 32 values, direct_stream_m
    wfree[5], wfree[6], wfree[7] = 0, 4_000_000_000, 75_000_000                  # a massless body (1/0 = inf: exactly 0), huge ones
    for w in (w1, (1 + np.arange(n) % 5).astype(np.uint32), wfree):
        ref64, norm, cpu32 = _refs(orc, pos, w, targets=tg)
        got = {}
        for mode in ("0", "1", "2", "3"):
            monkeypatch.setenv("NBODY_DIRECT_ASM", mode)
            ctx.set_params(arith=C.ARITH_AUTO, clamp=0.001)
            ctx.upload(pos, vel, w)
            got[mode] = ctx.accel_direct()
            check_fast(got[mode][tg], ref64, norm, cpu32, label=f" NBODY_DIRECT_ASM={mode}")
        assert np.array_equal(got["2"], got["3"])


# ------------------------------------------------------------------ the exact kernels' division (csrc/div_pair.h)
def test_div_pair_is_the_ieee_division_bit_for_bit(nb):
    """main.rs:252 divides a Vector2F by an f32: two IEEE divisions by one denominator.  The exact kernels run the compiler's
    own expansion of `/` for both quotients with the multiply-adds issued as packed instructions (csrc/div_pair.h); the
    quotients must be the host's IEEE quotients, bit for bit: random bit patterns (every exponent, both signs), denormal
    operands and results, zeros, infinities, overflow and underflow; NaN results need only be NaN on both sides."""
    rng = np.random.default_rng(52)
    n = 1 << 20

    def bits(k):
        return rng.integers(0, 1 << 32, k, dtype=np.uint64).astype(np.uint32).view(F32)

    def moderate(k):  # what the walks mostly see
        return (rng.standard_normal(k) * 10.0 ** rng.integers(-6, 7, k)).astype(F32)

    tiny, big = np.finfo(F32).tiny, np.finfo(F32).max
    special = np.array([0.0, -0.0, tiny, -tiny, tiny / 2, 1e-45, -1e-45, big, -big, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.0, 1.0 / 3.0,
                        2.0 ** -126, 2.0 ** -149, 2.0 ** 127, 1.5 * 2.0 ** -126, 0.75 * 2.0 ** -126], F32)
    g = np.meshgrid(special, special, special, indexing="ij")
    nx = np.concatenate([bits(n), moderate(n), g[0].ravel(), bits(n) * F32(0) + moderate(n)])
    ny = np.concatenate([bits(n), moderate(n), g[1].ravel(), bits(n)])
    den = np.concatenate([bits(n), np.abs(moderate(n)), g[2].ravel(), np.abs(moderate(n)) * F32(1e-30)])
    with np.errstate(all="ignore"):
        ex, ey = nx / den, ny / den
    qx, qy = nb._capi.selftest_div_pair(nx, ny, den)
    for got, want, name in ((qx, ex, "x"), (qy, ey, "y")):
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan), name
        bad = (got.view(np.uint32) != want.view(np.uint32)) & ~nan
        assert not bad.any(), (name, int(bad.sum()), nx[bad][:4], ny[bad][:4], den[bad][:4], got[bad][:4], want[bad][:4])
