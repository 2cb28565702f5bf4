"""A bounded, fixed-seed subset of tools/bvh_fuzz.py / tools/quad_fuzz.py inside the driver-run suite, against the ORACLE
(tools/*_fuzz.py compare the device builds with the product's own host builders over much larger sizes; here the checker is
the CPU restatement of bvh_tree.rs:56-158 / quad_tree.rs:153-270).  Needs an MI355X.

Every case: a random size (1 ... 2e5), leaf size and distribution; the device-built tree equals the oracle's node for node
(ranges, skip links, u32 masses, boxes and centres of gravity bit for bit, NaN for empty leaves) and so does the row
permutation; then two whole steps equal the oracle's World::update, row for row."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32
# Longer runs by hand (not the driver's): NBODY_FUZZ_CASES=200 NBODY_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py
CASES = int(os.environ.get("NBODY_FUZZ_CASES", "0"))
SEED = int(os.environ.get("NBODY_FUZZ_SEED", "0"))


def _scene(rng, kind, n, dtype, nb):
    if kind == 0:
        return (rng.random((n, 2)) * 1e5).astype(dtype)
    if kind == 1:
        return (rng.standard_normal((n, 2)) * 3e4).astype(dtype)          # both signs: sums wander through zero
    if kind == 2:
        return (-rng.random((n, 2)) * 1e5).astype(dtype)                   # all negative: the max fold starts from 0.0
    if kind == 3:
        return (10.0 ** rng.uniform(-6, 6, (n, 2))).astype(dtype)          # twelve decades
    if kind == 4:
        return (rng.integers(0, 3000, (n, 2)) * 0.5).astype(dtype)         # half-integer lattice: ties, coincident points
    return nb.scenes.plummer(n, seed=int(rng.integers(1, 1 << 30)), dtype=dtype)[0]


@pytest.fixture(scope="module")
def ctx(nb):
    c = nb._capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bvh_builds_equal_the_oracle_on_random_cases(nb, orc, ctx, dtype):
    C = nb._capi
    rng = np.random.default_rng(20261004 + (1 if dtype == np.float64 else 0) + 1000 * SEED)
    done = on_device = 0
    for case in range(CASES or (16 if dtype == np.float32 else 10)):
        n = int(10 ** rng.uniform(0.0, 5.3))
        leaf = int(rng.choice([1, 4, 16, 64, 64, 64, 200, 1000]))
        kind = int(rng.integers(0, 6))
        if leaf < 16:
            n = min(n, 30000)
        # (extended runs only: other thetas — long lists, the lane = target arm, the fused walk's back-off — on smaller scenes)
        theta = float(rng.choice([50.0, 50.0, 5.0, 0.7])) if CASES else 50.0
        if theta < 50.0:
            n = min(n, 15000)
        pos = _scene(rng, kind, n, dtype, nb)
        w = rng.integers(1, 9, n).astype(np.uint32)
        vel = (rng.standard_normal((n, 2)) * 10).astype(dtype)
        tag = f"case {case}: n {n} leaf {leaf} kind {kind} theta {theta} {np.dtype(dtype).name}"
        bvh = orc.BVH(pos, w, leaf_size=leaf)
        o = bvh.flat()
        if o.overflow:                                                      # > leaf coincident points: the reference recurses without end
            continue
        ctx.set_params(theta=theta, leaf_size=leaf, order=C.ORDER_AS_WRITTEN, arith=C.ARITH_AUTO)
        ctx.upload(pos, vel, w)
        ctx.accel_tree(C.TREE_BVH, pos[:1])
        on_device += int(ctx.last_build_on_device())
        t = ctx.tree_export()
        for k in ("mass", "is_leaf", "first", "count", "skip"):
            assert np.array_equal(t[k], getattr(o, k)), f"{tag}: {k}"
        assert np.array_equal(t["geom"], o.geom, equal_nan=True), f"{tag}: geom"
        assert np.array_equal(t["order"], o.ids), f"{tag}: permutation"
        try:
            rp, rv, rw, rids, _ = orc.update_bvh(pos, vel, w, delta=0.05, theta=theta, leaf_size=leaf, mode=orc.AS_WRITTEN, nsteps=2, nthreads=16)
        except RuntimeError:                                                # points that come to coincide during the steps
            continue
        ctx.upload(pos, vel, w)
        ctx.update_tree(C.TREE_BVH, 0.05, 2)
        p, v, w2, ids = ctx.download()
        assert np.array_equal(ids, rids) and np.array_equal(p, rp, equal_nan=True) and np.array_equal(v, rv, equal_nan=True) and np.array_equal(w2, rw), tag
        done += 1
    assert done >= 6 and on_device >= done // 2, (done, on_device)
    ctx.set_params(leaf_size=64, theta=50.0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quad_builds_equal_the_oracle_on_random_cases(nb, orc, ctx, dtype):
    C = nb._capi
    rng = np.random.default_rng(20261005 + (1 if dtype == np.float64 else 0) + 1000 * SEED)
    done = 0
    for case in range(CASES or (14 if dtype == np.float32 else 8)):
        n = int(10 ** rng.uniform(0.0, 5.3))
        kind = int(rng.choice([0, 1, 3, 4, 5, 5]))                          # (kind 1 and 3 put points outside the root cell too)
        pos = _scene(rng, kind, n, dtype, nb)
        if kind == 4:
            pos = pos + dtype(0.25) * rng.integers(0, 3, pos.shape).astype(dtype)   # fewer than nine coincident points per site
        w = rng.integers(1, 9, n).astype(np.uint32)
        vel = (rng.standard_normal((n, 2)) * 10).astype(dtype)
        tag = f"case {case}: n {n} kind {kind} {np.dtype(dtype).name}"
        quad = orc.Quad(pos, w)
        o = quad.flat()
        if o.overflow:
            continue
        ctx.set_params(theta=0.7, order=C.ORDER_CONSISTENT, arith=C.ARITH_AUTO)
        ctx.upload(pos, vel, w)
        ctx.accel_tree(C.TREE_QUAD, pos[:1])
        t = ctx.tree_export()
        for k in ("mass", "is_leaf", "first", "count", "skip"):
            assert np.array_equal(t[k], getattr(o, k)), f"{tag}: {k}"
        assert np.array_equal(t["geom"], o.geom, equal_nan=True), f"{tag}: geom"
        assert np.array_equal(t["order"], o.order), f"{tag}: leaf order"
        try:
            rp, rv, _ = orc.update_quad(pos, vel, w, delta=0.05, theta=0.7, nsteps=2, nthreads=16)
        except RuntimeError:
            continue
        ctx.upload(pos, vel, w)
        ctx.update_tree(C.TREE_QUAD, 0.05, 2)
        p, v, _, ids = ctx.download()
        assert np.array_equal(ids, np.arange(n)) and np.array_equal(p, rp, equal_nan=True) and np.array_equal(v, rv, equal_nan=True), tag
        done += 1
    assert done >= 5, done


def test_direct_main_pass_on_random_cases(nb, orc, lab_ctx, monkeypatch):
    """The direct step's packed / streamed main pass (NBODY_DIRECT_ASM 2 / 3) on random sizes (the near/far split on), random mass
    patterns (equal, a few classes, a few heavy bodies, all different) and positions with coincident bodies and pairs inside the
    clamp radius: within the oracle's tolerance on sampled targets, bitwise reproducible, and 2 and 3 give the same bits."""
    from tests._tol import check_fast
    C = nb._capi
    ctx = lab_ctx                                            # laboratory library: NBODY_DIRECT_ASM is one of its switches
    rng = np.random.default_rng(20261006 + 1000 * SEED)
    for case in range(CASES // 10 or 3):
        n = int(rng.integers(65536, 180000))
        kind = int(rng.integers(0, 3))
        pos = _scene(rng, (0, 1, 5)[kind], n, np.float32, nb)
        dup = rng.integers(0, n, 8)
        pos[dup[:4]] = pos[dup[4:]]                                       # coincident bodies
        near = rng.integers(0, n, 16)
        pos[near[:8]] = pos[near[8:]] + F32(0.0078125)                   # inside the clamp radius
        mk = int(rng.integers(0, 4))
        if mk == 0:
            w = np.full(n, int(rng.integers(1, 1000)), np.uint32)
        elif mk == 1:
            w = rng.integers(1, 6, n).astype(np.uint32)
        elif mk == 2:
            w = np.ones(n, np.uint32)
            w[rng.integers(0, n, 5)] = rng.integers(1000, 1 << 26, 5)
        else:
            w = rng.integers(1, 1 << 20, n).astype(np.uint32)
        vel = np.zeros_like(pos)
        tg = rng.integers(0, n, 1024)
        ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
        cpu32, _ = orc.direct_accel(pos, w, targets=tg, nthreads=16)
        got = {}
        for mode in ("2", "3"):
            monkeypatch.setenv("NBODY_DIRECT_ASM", mode)
            ctx.set_params(arith=C.ARITH_AUTO, clamp=0.001, theta=50.0, leaf_size=64)
            ctx.upload(pos, vel, w)
            got[mode] = ctx.accel_direct()
            check_fast(got[mode][tg], ref64, norm, cpu32, label=f" case {case}: n {n} scene {kind} masses {mk} NBODY_DIRECT_ASM={mode}")
            ctx.upload(pos, vel, w)
            assert np.array_equal(ctx.accel_direct(), got[mode])
        assert np.array_equal(got["2"], got["3"]), f"case {case}: n {n} scene {kind} masses {mk}"


def test_fast_walks_on_random_cases(nb, orc, ctx, monkeypatch):
    """The tolerance-contract walks (walk_tile_fast; the FAST arm of the quad walk) on random sizes, leaf sizes, thetas and
    distributions: every target within 2e-5 of the sum of its terms' magnitudes against the oracle's walk_ref (the walk's own
    interaction list, terms as main.rs:252 writes them, summed in double)."""
    from tests._tol import check_fast
    C = nb._capi
    monkeypatch.setenv("NBODY_WALK_SPLIT", "3")            # the one-pass walk whatever the size
    rng = np.random.default_rng(20261007 + 1000 * SEED)
    done = 0
    for case in range(CASES // 4 or 6):
        n = int(10 ** rng.uniform(3.0, 5.2))
        leaf = int(rng.choice([16, 64, 64, 200]))
        theta = float(rng.choice([50.0, 50.0, 5.0, 0.7]))
        if theta < 50.0:
            n = min(n, 20000)
        kind = int(rng.integers(0, 6))
        pos = _scene(rng, kind, n, np.float32, nb)
        if kind == 4:
            pos = pos + F32(0.25) * rng.integers(0, 3, pos.shape).astype(F32)
        w = rng.integers(1, 9, n).astype(np.uint32)
        vel = np.zeros_like(pos)
        tag = f" case {case}: n {n} leaf {leaf} theta {theta} kind {kind}"
        bvh = orc.BVH(pos, w, leaf_size=leaf)
        flat = bvh.flat()
        if not flat.overflow:
            ctx.set_params(theta=theta, leaf_size=leaf, order=C.ORDER_CONSISTENT, arith=C.ARITH_FAST)
            ctx.upload(pos, vel, w)
            acc = ctx.accel_tree(C.TREE_BVH)
            ref64, norm = bvh.walk_ref(flat.pos_perm, theta=theta, nthreads=16)
            check_fast(acc, ref64, np.maximum(norm, 1e-300), label=" bvh" + tag)
            done += 1
        quad = orc.Quad(pos, w)
        qtheta = float(rng.choice([5.0, 0.7, 0.25, 0.0]))                  # 0: the direct sum in disguise (lists of n terms)
        if not quad.flat().overflow and n <= (60000 if qtheta >= 0.7 else 30000):
            ctx.set_params(theta=qtheta, order=C.ORDER_CONSISTENT, arith=C.ARITH_FAST)
            ctx.upload(pos, vel, w)
            acc = ctx.accel_tree(C.TREE_QUAD)
            ref64, norm = quad.walk_ref(pos, theta=qtheta, nthreads=16)
            check_fast(acc, ref64, np.maximum(norm, 1e-300), label=f" quad theta {qtheta}" + tag)
    assert done >= 3
    ctx.set_params(arith=C.ARITH_AUTO, leaf_size=64, theta=50.0, order=C.ORDER_AS_WRITTEN)


def test_sharded_direct_steps_on_random_cases(nb, orc):
    """Several ranks behind one handle (the one device listed several times, peer copies), random sizes with the near/far split
    on (the streamed main pass over couples, ragged last blocks, 1-4 chunks; equal masses, mass classes or free per-body masses): one FAST step from
    rest gives v = fl(a dt) with a inside the frozen tolerance on sampled targets, and x = x + v dt bit for bit."""
    from tests._tol import ACC_RTOL
    C = nb._capi
    rng = np.random.default_rng(20261008 + 1000 * SEED)
    for case in range(CASES // 20 or 2):
        n = int(rng.integers(66000, 150000))
        ranks = int(rng.choice([2, 3, 4, 8]))
        chunks = int(rng.choice([1, 2, 4]))
        pos = _scene(rng, (0, 1, 5)[int(rng.integers(0, 3))], n, np.float32, nb)
        mk = int(rng.integers(0, 3))                             # equal masses | four mass classes | free masses (direct_stream_m)
        w = (np.ones(n, np.uint32), rng.integers(1, 5, n).astype(np.uint32), rng.integers(1, 1 << 20, n).astype(np.uint32))[mk]
        if case == 0:
            w = rng.integers(1, 1 << 20, n).astype(np.uint32)    # every run covers the free-mass path at least once
        vel = np.zeros_like(pos)
        tg = np.sort(rng.choice(n, 2048, replace=False))
        ref64, norm = orc.direct_accel(pos, w, targets=tg, accum="f64", nthreads=16)
        m = C.MultiContext([0] * ranks, C.EXCHANGE_PEER, chunks)
        try:
            m.set_params(arith=C.ARITH_FAST)
            m.upload(pos, vel, w)
            m.update_direct(0.1, 1)
            p, v, _, ids = m.download()
        finally:
            m.close()
        tag = f"case {case}: n {n} ranks {ranks} chunks {chunks}"
        assert np.array_equal(ids, np.arange(n)), tag
        err = np.abs(v[tg].astype(np.float64) / 0.1 - ref64).sum(axis=1)
        slack = 4 * np.finfo(F32).eps * np.abs(ref64).sum(axis=1)
        assert np.all(err <= ACC_RTOL * norm + slack), (tag, float(((err - slack) / norm).max()))
        assert np.array_equal(p, (pos + v * F32(0.1)).astype(F32)), tag
