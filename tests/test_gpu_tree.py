"""Parity of the Barnes-Hut HIP path (host build + device walk) against the CPU oracle.  Needs an MI355X.

The walk uses the reference's operations in the reference's DFS order, so everything here is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def ctx(nb):
    c = nb._capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,theta", [(40, 50.0), (1024, 50.0), (1024, 0.5), (5000, 0.5), (5000, 0.0), (30000, 1.0)])
def test_bvh_walk_bit_exact(nb, orc, ctx, dtype, n, theta):
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(n, seed=51)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    w = (np.arange(n) % 6 + 1).astype(np.uint32)
    ctx.set_params(theta=theta, leaf_size=64)
    ctx.upload(pos, vel, w)
    # arbitrary targets (the snapshot positions, in upload order)
    acc = ctx.accel_tree(C.TREE_BVH, pos)
    bvh = orc.BVH(pos, w)
    ref = bvh.walk(pos, theta=theta, nthreads=8)
    assert np.array_equal(acc, ref)
    # tree exported by the library == oracle tree; rows were permuted as BVHTree::from permutes self.particles
    t, o = ctx.tree_export(), bvh.flat()
    for k in ("mass", "is_leaf", "first", "count", "skip"):
        assert np.array_equal(t[k], getattr(o, k)), k
    assert np.array_equal(t["geom"], o.geom, equal_nan=True)
    p, v, w2, ids = ctx.download()
    assert np.array_equal(ids, o.ids) and np.array_equal(p, o.pos_perm) and np.array_equal(w2, w[o.ids])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,theta", [(5, 0.5), (1024, 0.5), (6000, 0.5), (6000, 0.0), (20000, 50.0)])
def test_quad_walk_bit_exact(nb, orc, ctx, dtype, n, theta):
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(n, seed=52)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    ctx.set_params(theta=theta)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_tree(C.TREE_QUAD)          # the particles themselves
    ref = orc.Quad(pos, w).walk(pos, theta=theta, nthreads=8)
    assert np.array_equal(acc, ref)
    acc2 = ctx.accel_tree(C.TREE_QUAD, pos[::7])
    assert np.array_equal(acc2, ref[::7])
    p, _, _, ids = ctx.download()
    assert np.array_equal(ids, np.arange(n)) and np.array_equal(p, pos)   # the quad build does not permute rows


@pytest.mark.parametrize("order", ["as_written", "consistent"])
@pytest.mark.parametrize("theta", [50.0, 0.5])
def test_update_bvh_trajectory_bit_exact(nb, orc, order, theta):
    """World::update over 20 steps, both application orders, against the restatement."""
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    world = nb.World(pos, vel, w, method="bvh", theta=theta, order=order)
    cnt = nb.Counting()
    for _ in range(20):
        world.update(0.1, cnt)
    p, v, w2, ids = world.particles()
    world.close()
    mode = orc.AS_WRITTEN if order == "as_written" else orc.CONSISTENT
    rp, rv, rw, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=theta, mode=mode, nsteps=20, nthreads=8)
    assert np.array_equal(ids, rids)
    assert np.array_equal(p, rp) and np.array_equal(v, rv) and np.array_equal(w2, rw)
    assert cnt.build_bvh > 0 and cnt.sum_gravity > 0 and cnt.post_calculations > 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_update_quad_trajectory_bit_exact(nb, orc, dtype):
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    world = nb.World(pos, vel, w, method="quad", theta=0.5)
    world.update(0.1, None, n_steps=20)
    p, v, _, ids = world.particles()
    world.close()
    rp, rv, _ = orc.update_quad(pos, vel, w, delta=0.1, theta=0.5, nsteps=20, nthreads=8)
    assert np.array_equal(p, rp) and np.array_equal(v, rv)


def test_update_bvh_f64(nb, orc):
    pos, vel, w = nb.scenes.plummer(4096, seed=53, dtype=np.float64)
    world = nb.World(pos, vel, w, method="bvh", theta=0.5, order="consistent")
    world.update(0.1, None, n_steps=5)
    p, v, _, ids = world.particles()
    world.close()
    rp, rv, _, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=0.5, mode=orc.CONSISTENT, nsteps=5, nthreads=8)
    assert np.array_equal(ids, rids) and np.array_equal(p, rp) and np.array_equal(v, rv)


def test_walk_statistics_match_oracle_counts(nb, orc, ctx):
    """Node visits / accepted nodes / leaf pairs (the algorithmic-bytes basis of the walk's roofline)."""
    C = nb._capi
    n = 20000
    pos, vel, w = nb.scenes.plummer(n, seed=54)
    ctx.set_params(theta=0.5)
    ctx.upload(pos, vel, w)
    ctx.walk_stats(True)
    ctx.accel_tree(C.TREE_BVH, pos[:3000])
    got = ctx.walk_stats(False)
    _, st = orc.BVH(pos, w).walk(pos[:3000], theta=0.5, stats=True)
    assert tuple(int(x) for x in st) == got


def test_degenerate_input_is_an_error_not_a_hang(nb, ctx):
    C = nb._capi
    pos = np.tile(np.array([[5.0, 5.0]], F32), (200, 1))
    ctx.upload(pos, np.zeros_like(pos), None)
    with pytest.raises(C.NBodyError) as e:
        ctx.update_tree(C.TREE_BVH, 0.1, 1)
    assert e.value.code == C.ERR_DEGENERATE


def test_config4_4M_quad_f64_sampled(nb, orc):
    """BASELINE config 4: 4 194 304 bodies, Barnes-Hut theta = 0.5 over the linearised quad tree, f64.
    Sampled targets are compared bit for bit with the CPU restatement; one full step runs on the device."""
    C = nb._capi
    n = 1 << 22
    pos, vel, w = nb.scenes.plummer(n, seed=0x5EED0004, dtype=np.float64)
    c = C.Context(0)
    try:
        c.set_params(theta=0.5)
        c.upload(pos, vel, w)
        tg = pos[::1024]
        acc = c.accel_tree(C.TREE_QUAD, tg)
        ref = orc.Quad(pos, w).walk(tg, theta=0.5, nthreads=16)
        assert np.array_equal(acc, ref)
        cnt = C.Counting()
        c.update_tree(C.TREE_QUAD, 0.1, 1, cnt)
        p, v, _, _ = c.download()
        assert np.all(np.isfinite(p)) and np.all(np.isfinite(v))
        print(f"4M quad f64 step: build {cnt.build_bvh:.3f}s walk {cnt.sum_gravity:.3f}s integrate {cnt.post_calculations:.4f}s")
    finally:
        c.close()


# ------------------------------------------------------------------ device-side quad build (quad_build.hip)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 8, 9, 10, 100, 5000, 200000])
def test_device_quad_build_equals_oracle_tree(nb, orc, ctx, dtype, n):
    """The tree the device builds (path keys + two radix sorts + scans) must be the reference's tree: every cell,
    child order, leaf slot order, mass and centre of gravity, bit for bit."""
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(n, seed=91, dtype=dtype)
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    ctx.set_params(theta=0.5)
    ctx.upload(pos, vel, w)
    acc = ctx.accel_tree(C.TREE_QUAD)
    t = ctx.tree_export()
    o = orc.Quad(pos, w).flat()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(t[k], getattr(o, k)), k
    assert t["max_depth"] == int(o.depth.max())
    assert np.array_equal(acc, orc.Quad(pos, w).walk(pos, theta=0.5, nthreads=8))


def test_device_and_host_quad_builds_agree(nb, ctx, monkeypatch):
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()          # lattice points: many equal coordinates, mixed masses
    ctx.set_params(theta=0.5)
    ctx.upload(pos, vel, w)
    ctx.accel_tree(C.TREE_QUAD, pos[:10])
    dev = ctx.tree_export()
    monkeypatch.setenv("NBODY_TREE_BUILD_HOST", "1")
    ctx.accel_tree(C.TREE_QUAD, pos[:10])
    host = ctx.tree_export()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(dev[k], host[k]), k


def test_device_quad_build_declines_deep_trees_and_host_takes_over(nb, orc, ctx):
    """9 points within 1e-8 of each other need ~43 levels in f64: beyond the 31 levels of the device's path key.
    The step must still be exact (host builder) rather than wrong or an error."""
    C = nb._capi
    rng = np.random.default_rng(7)
    pos = (rng.random((2000, 2)) * 1e5)
    pos[100:109] = pos[100] + rng.random((9, 2)) * 1e-8
    w = np.ones(2000, np.uint32)
    ctx.set_params(theta=0.5)
    ctx.upload(pos, np.zeros_like(pos), w)
    acc = ctx.accel_tree(C.TREE_QUAD)
    o = orc.Quad(pos, w)
    assert int(o.flat().depth.max()) > 31
    assert ctx.tree_export()["max_depth"] == int(o.flat().depth.max())
    assert np.array_equal(acc, o.walk(pos, theta=0.5, nthreads=8))


def test_device_quad_points_outside_root_cell(nb, orc, ctx):
    C = nb._capi
    rng = np.random.default_rng(5)
    inside = (rng.random((3000, 2)) * 1e5).astype(F32)
    outside = np.array([[-5e4, 2e4], [1.7e5, 3e4], [4e4, -1e3], [5e4, 2.5e5], [-1.0, -1.0], [1e5 + 1, 1e5 + 1]], F32)
    pos = np.concatenate([inside[:1500], outside, inside[1500:]])
    ctx.set_params(theta=0.5)
    ctx.upload(pos, np.zeros_like(pos), None)
    ctx.accel_tree(C.TREE_QUAD, pos[:4])
    t = ctx.tree_export()
    o = orc.Quad(pos).flat()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(t[k], getattr(o, k)), k


def test_cpp_world_mirror_headless_driver(nb):
    """csrc/world.hpp + csrc/nbody_run.cpp: the C++ host mirror of World::update over the C ABI runs the reference's
    scene and prints the reference's once-a-second block (main.rs:149-156)."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(nb._capi.LIB_PATH), "nbody_run")
    if not os.path.exists(exe):
        pytest.skip("nbody_run not built")
    r = subprocess.run([exe, "5", "bvh", "12345"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = re.search(r"len: (\d+)", r.stdout)
    assert m and 140000 < int(m.group(1)) < 165000          # 2 heavy + ~51k lattice + 100 000 (main.rs:343)
    assert "step: 5" in r.stdout and "Counting { build_bvh:" in r.stdout
    assert "centroid after 5 steps" in r.stdout


@pytest.mark.parametrize("dtype,kind", [(np.float32, "bvh"), (np.float32, "quad"), (np.float64, "quad")])
def test_fast_walk_is_within_tolerance_of_the_exact_walk(nb, orc, ctx, dtype, kind):
    """arith = FAST on a tree: same node tests and interaction lists, one reciprocal per pair.  Not bit parity: the
    per-target error must stay within 2e-5 of sum|term| (f32) / 1e-12 (f64) of the reference walk."""
    C = nb._capi
    n = 20000
    pos, vel, _ = nb.scenes.plummer(n, seed=93, dtype=dtype)
    pos[10] = pos[11]                                   # a coincident pair inside one leaf
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    k = C.TREE_BVH if kind == "bvh" else C.TREE_QUAD
    ctx.set_params(theta=0.5, arith=C.ARITH_FAST)
    ctx.upload(pos, vel, w)
    fast = ctx.accel_tree(k, pos)
    ctx.set_params(arith=C.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    exact = ctx.accel_tree(k, pos)
    tree = orc.BVH(pos, w) if kind == "bvh" else orc.Quad(pos, w)
    assert np.array_equal(exact, tree.walk(pos, theta=0.5, nthreads=8))
    assert np.all(np.isfinite(fast)) and not np.array_equal(fast, exact)
    # scale: the direct-sum norm bounds the walk's sum of |term| from above for the particle part; use |a| + norm/N
    _, norm = orc.direct_accel(pos.astype(np.float32), w, targets=np.arange(0, n, 50), accum="f64", nthreads=8)
    err = np.abs(fast[::50].astype(np.float64) - exact[::50]).sum(axis=1)
    tol = 2e-5 if dtype == np.float32 else 1e-12
    assert np.all(err <= tol * norm), float((err / norm).max())


# ------------------------------------------------------------------ device-side BVH build (bvh_build.hip)
def _bvh_export_equal(a, b):
    for k in ("mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["geom"], b["geom"], equal_nan=True)
    assert a["max_depth"] == b["max_depth"]


def _bvh_scenes(nb):
    rng = np.random.default_rng(17)
    out = {}
    for n in (1, 2, 3, 64, 65, 66, 129, 1000, 4097, 20000, 300000):
        p, _, _ = nb.scenes.plummer(n, seed=300 + n)
        out[f"plummer{n}"] = (p, (np.arange(n) % 7 + 1).astype(np.uint32))
    g = nb.scenes.galaxy()
    out["galaxy"] = (g[0], g[2])
    # coordinates of both signs: the running sums wander through zero and many binades
    out["centred"] = ((rng.standard_normal((50000, 2)) * 3e4).astype(F32), np.ones(50000, np.uint32))
    # everything negative: the max fold starts from 0.0 (bvh_tree.rs:59) and the chain carries a sign
    out["negative"] = ((-rng.random((30000, 2)) * 1e5).astype(F32), np.ones(30000, np.uint32))
    # half-integers on a small lattice: ties in almost every add, thousands of equal coordinates
    out["ties"] = ((rng.integers(0, 400, (40000, 2)) * 0.5).astype(F32), np.ones(40000, np.uint32))
    # 16 decades of magnitudes
    out["wide"] = ((10.0 ** rng.uniform(-8, 8, (20000, 2))).astype(F32), np.ones(20000, np.uint32))
    # u32 masses that wrap
    out["wrap"] = ((rng.random((5000, 2)) * 1e5).astype(F32), np.full(5000, 0x7FFFFFFF, np.uint32))
    return out


@pytest.mark.parametrize("leaf", [64, 8, 1, 200, 5000])
def test_device_bvh_build_equals_oracle_tree(nb, orc, ctx, leaf):
    """The tree built on the device (exact-sum scan, rank-list partition, path-key numbering) is the reference's tree:
    boxes, split order, leaf ranges, permutation, masses and centres of gravity, bit for bit."""
    C = nb._capi
    for name, (pos, w) in _bvh_scenes(nb).items():
        if leaf < 8 and pos.shape[0] > 50000:
            continue
        prm = C.default_params()
        prm.leaf_size = leaf
        if C.host_tree(C.TREE_BVH, pos, w, prm)["overflow"]:
            continue   # the reference itself recurses without end here (coincident points, or lattice columns whose
                       # mean no point exceeds): an error by definition, covered by the degenerate-input tests
        ctx.set_params(theta=50.0, leaf_size=leaf)
        ctx.upload(pos, np.zeros_like(pos), w)
        ctx.accel_tree(C.TREE_BVH, pos[:4])
        assert ctx.last_build_on_device(), name
        t = ctx.tree_export()
        o = orc.BVH(pos, w, leaf_size=leaf).flat()
        for k in ("mass", "is_leaf", "first", "count", "skip"):
            assert np.array_equal(t[k], getattr(o, k)), (name, k)
        assert np.array_equal(t["geom"], o.geom, equal_nan=True), name
        assert np.array_equal(t["order"], o.ids), name
        p, _, w2, ids = ctx.download()
        assert np.array_equal(ids, o.ids) and np.array_equal(p, o.pos_perm) and np.array_equal(w2, w[o.ids]), name


@pytest.mark.parametrize("scene", ["plummer_1m", "centred_1m5", "negative_700k"])
def test_device_bvh_build_equals_the_host_builder_on_long_chains(nb, ctx, scene):
    """The same at sizes where the chain is folded out of LDS windows over many rounds, with prepared chunk runs, crossings
    seen coming and same-sign rounds proven by their end (bvh_build.hip exact_fold): against the host builder, which adds one
    point after the other (tree_build.hpp; itself checked against the oracle at the smaller sizes above)."""
    C = nb._capi
    rng = np.random.default_rng(23)
    if scene == "plummer_1m":
        pos = nb.scenes.plummer(1 << 20, seed=0x5EED0003)[0]
    elif scene == "centred_1m5":
        pos = (rng.standard_normal((1_500_000, 2)) * 3e4).astype(F32)   # sums wander through zero: mixed-sign rounds
    else:
        pos = (-rng.random((700_000, 2)) * 1e5).astype(F32)             # the chain carries a sign
    n = pos.shape[0]
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    prm = C.default_params()
    h = C.host_tree(C.TREE_BVH, pos, w, prm)
    assert not h["overflow"]
    ctx.set_params(theta=50.0, leaf_size=prm.leaf_size)  # (the shared context keeps whatever the test before it set)
    ctx.upload(pos, np.zeros_like(pos), w)
    ctx.accel_tree(C.TREE_BVH, pos[:4])
    assert ctx.last_build_on_device(), scene
    _bvh_export_equal(ctx.tree_export(), h)


def test_device_and_host_bvh_builds_agree_over_steps(nb, monkeypatch):
    """20 full steps of the reference scene with the device build against 20 with the host build: same rows."""
    pos, vel, w = nb.scenes.galaxy()
    res = []
    for host in ("0", "1"):
        monkeypatch.setenv("NBODY_TREE_BUILD_HOST", host)
        world = nb.World(pos, vel, w, method="bvh")
        cnt = nb.Counting()
        for _ in range(20):
            world.update(0.1, cnt)
        assert world.ctx.last_build_on_device() == (host == "0")
        res.append(world.particles())
        world.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_device_bvh_build_then_host_build_keeps_weights_in_row_order(nb, orc, ctx, monkeypatch):
    """A host build after device builds has to see the weights in the permuted row order."""
    C = nb._capi
    pos, vel, _ = nb.scenes.plummer(3000, seed=77)
    w = (np.arange(3000) % 11 + 1).astype(np.uint32)
    ctx.set_params(theta=0.5, leaf_size=64)
    ctx.upload(pos, vel, w)
    ctx.update_tree(C.TREE_BVH, 0.1, 2)
    assert ctx.last_build_on_device()
    monkeypatch.setenv("NBODY_TREE_BUILD_HOST", "1")
    ctx.update_tree(C.TREE_BVH, 0.1, 1)
    assert not ctx.last_build_on_device()
    p, v, w2, ids = ctx.download()
    rp, rv, rw, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=0.5, mode=orc.AS_WRITTEN, nsteps=3, nthreads=8)
    assert np.array_equal(ids, rids) and np.array_equal(p, rp) and np.array_equal(v, rv) and np.array_equal(w2, rw)


def test_device_bvh_build_declines_nan_positions(nb, orc, ctx):
    """minps/maxps are order-dependent on NaN: the device build hands such input to the host builder."""
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(2000, seed=5)
    pos = pos.copy()
    pos[777, 0] = np.nan
    ctx.set_params(theta=50.0, leaf_size=64)
    ctx.upload(pos, vel, w)
    ctx.accel_tree(C.TREE_BVH, pos[:4])
    assert not ctx.last_build_on_device()
    o = orc.BVH(pos, w).flat()
    t = ctx.tree_export()
    assert np.array_equal(t["geom"], o.geom, equal_nan=True) and np.array_equal(t["order"], o.ids)


@pytest.mark.parametrize("blind", ["1", "3"])
def test_device_bvh_build_with_long_nodes_left_after_the_blind_levels(nb, orc, lab_ctx, monkeypatch, blind):
    """The build enqueues the levels a balanced tree needs and checks once; a lopsided tree (or, here, too few blind
    levels) must go on level by level and still end in the same tree."""
    C = nb._capi
    ctx = lab_ctx                                          # laboratory library: the test forces a variant / a slow path
    monkeypatch.setenv("NBODY_BVH_BLIND_LEVELS", blind)
    rng = np.random.default_rng(23)
    scenes = {"galaxy": nb.scenes.galaxy()[::2],
              "wide": ((10.0 ** rng.uniform(-8, 8, (60000, 2))).astype(F32), np.ones(60000, np.uint32))}
    for name, (pos, w) in scenes.items():
        ctx.set_params(theta=50.0, leaf_size=64)
        ctx.upload(pos, np.zeros_like(pos), w)
        ctx.accel_tree(C.TREE_BVH, pos[:4])
        assert ctx.last_build_on_device(), name
        t = ctx.tree_export()
        o = orc.BVH(pos, w).flat()
        for k in ("mass", "is_leaf", "first", "count", "skip"):
            assert np.array_equal(t[k], getattr(o, k)), (name, k)
        assert np.array_equal(t["geom"], o.geom, equal_nan=True), name
        assert np.array_equal(t["order"], o.ids), name


# ------------------------------------------------------------------ split walk (walk_split.hip)
@pytest.mark.parametrize("mode", ["0", "2", "3"])
def test_split_and_fused_walks_are_the_same_walk(nb, orc, lab_ctx, monkeypatch, mode):
    """count / emit / ordered-sum (NBODY_WALK_SPLIT=2), the one-pass walk with the terms through LDS (=3) and the fused
    wave walk (=0) against the CPU recursion: the
    same nodes, pairs, operations and order of additions, so the same bits — targets = the particles (tree order), a
    strided subset of arbitrary targets, coincident and out-of-box targets, several thetas."""
    C = nb._capi
    ctx = lab_ctx                                          # laboratory library: the test forces a variant / a slow path
    monkeypatch.setenv("NBODY_WALK_SPLIT", mode)
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::3].copy(), vel[::3].copy(), w[::3].copy()
    pos[100] = pos[101]                                    # a pair the reference skips (sum == 0)
    bvh = orc.BVH(pos, w)
    extra = np.array([[-5e4, 2e5], [6e4, 6e4], [0, 0], [1e9, -1e9]], F32)
    for theta in (50.0, 2.0):
        ctx.set_params(theta=theta, leaf_size=64, order=C.ORDER_CONSISTENT)
        ctx.upload(pos, vel, w)
        assert np.array_equal(ctx.accel_tree(C.TREE_BVH), bvh.walk(bvh.flat().pos_perm, theta=theta, nthreads=8))
        tg = np.concatenate([pos[::7], extra])
        ctx.upload(pos, vel, w)                            # a build permutes the rows, and the tree depends on their order
        assert np.array_equal(ctx.accel_tree(C.TREE_BVH, tg), bvh.walk(tg, theta=theta, nthreads=8))


def test_split_walk_steps_equal_fused_walk_steps(nb, lab, monkeypatch):
    pos, vel, w = nb.scenes.galaxy()
    res = []
    for mode in ("0", "2", "3", "1"):
        monkeypatch.setenv("NBODY_WALK_SPLIT", mode)
        world = nb.World(pos, vel, w, method="bvh")
        cnt = nb.Counting()
        for _ in range(5):
            world.update(0.1, cnt)
        res.append(world.particles())
        world.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)


def test_split_walk_backs_off_when_the_terms_do_not_fit(nb, lab, monkeypatch):
    """N = 2^21 on the reference's needle-box BVH needs more terms than 32-bit offsets hold: the three-pass walk must fall
    back to the fused walk, the one-pass walk must go on without a counted estimate, and both must give the same
    accelerations as the fused walk."""
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(1 << 21, seed=0x5EED0009)
    out = []
    for mode in ("1", "4", "0"):
        monkeypatch.setenv("NBODY_WALK_SPLIT", mode)
        with C.Context(0) as c:
            c.set_params(theta=50.0)
            c.upload(pos, vel, w)
            out.append(c.accel_tree(C.TREE_BVH))
    assert np.array_equal(out[0], out[2]) and np.array_equal(out[1], out[2])


def test_device_quad_build_sorts_by_the_previous_depth_and_recovers_when_the_tree_got_deeper(nb, orc):
    """The quad build's radix sorts look at (depth of the previous tree + 3) levels only; a tree that turns out deeper has
    to be noticed and built again with all levels, bit for bit the same as ever."""
    C = nb._capi
    rng = np.random.default_rng(77)
    n = 40000
    shallow = (rng.random((n, 2)) * 1e5).astype(F32)                 # depth ~9
    deep = shallow.copy()
    deep[:3000] = (5e4 + rng.random((3000, 2)) * 50.0).astype(F32)   # 3000 points in a 50-unit box: ~8 levels more
    w = np.ones(n, np.uint32)
    with C.Context(0) as ctx:
        ctx.set_params(theta=0.5)
        for pos in (shallow, deep, shallow):
            ctx.upload(pos, np.zeros_like(pos), w)
            ctx.accel_tree(C.TREE_QUAD, pos[:8])
            assert ctx.last_build_on_device()
            t, o = ctx.tree_export(), orc.Quad(pos, w).flat()
            for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
                assert np.array_equal(t[k], getattr(o, k)), k
            assert t["max_depth"] == int(o.depth.max())
        d_shallow, d_deep = int(orc.Quad(shallow, w).flat().depth.max()), int(orc.Quad(deep, w).flat().depth.max())
        assert d_deep > d_shallow + 3


# ------------------------------------------------------------------ leaf sizes through the two lane-dealing walks
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("leaf", [1, 3, 8, 15, 16, 40, 64, 100, 200])
def test_bvh_walk_bit_exact_for_every_leaf_size(nb, orc, ctx, monkeypatch, dtype, leaf):
    """leaf_size < 16 takes tree_walk_small (pairs dealt to the lanes, rounds of floor(64 / m) targets), >= 16 in f32 the
    one-pass walk with the LDS tile (leaves longer than 64 particles go through it 64 at a time); f64 big leaves the fused
    walk.  Particles themselves and arbitrary targets (fewer than a wave, not a multiple of 64, outside every box)."""
    C = nb._capi
    monkeypatch.setenv("NBODY_WALK_SPLIT", "3")
    n = 9000
    pos, vel, _ = nb.scenes.plummer(n, seed=61)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    if leaf >= 2:
        pos[200] = pos[201]                               # a pair the reference skips
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    prm = C.default_params()
    prm.leaf_size = leaf
    if C.host_tree(C.TREE_BVH, pos, w, prm)["overflow"]:
        pytest.skip("degenerate for this leaf size")
    bvh = orc.BVH(pos, w, leaf_size=leaf)
    for theta in (50.0, 3.0):
        ctx.set_params(theta=theta, leaf_size=leaf, order=C.ORDER_CONSISTENT)
        ctx.upload(pos, vel, w)
        assert np.array_equal(ctx.accel_tree(C.TREE_BVH), bvh.walk(bvh.flat().pos_perm, theta=theta, nthreads=8))
        for tg in (pos[:37], np.concatenate([pos[5::11], np.array([[1e7, -1e7], [0, 0]], dtype)])):
            ctx.upload(pos, vel, w)
            assert np.array_equal(ctx.accel_tree(C.TREE_BVH, tg), bvh.walk(tg, theta=theta, nthreads=8))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quad_walk_with_full_and_sparse_leaves(nb, orc, ctx, dtype):
    """tree_walk_small on leaves of every size 1..8 (clumps of k points far apart) and on a lattice whose leaves are
    all full, theta 0.5 and 0 (everything opened)."""
    C = nb._capi
    rng = np.random.default_rng(8)
    clumps = []
    for k in range(1, 9):
        for _ in range(40):
            centre = rng.random(2) * 9e4 + 5e3
            clumps.append(centre + rng.random((k, 2)) * 0.5)
    sparse = np.concatenate(clumps).astype(dtype)
    g = np.arange(64, dtype=np.float64) * 1500.0 + 700.0
    lattice = np.stack(np.meshgrid(g, g), -1).reshape(-1, 2)
    lattice = (lattice + rng.random(lattice.shape) * 3.0).astype(dtype)
    for pos in (sparse, lattice):
        w = (np.arange(pos.shape[0]) % 4 + 1).astype(np.uint32)
        for theta in (0.5, 0.0):
            ctx.set_params(theta=theta)
            ctx.upload(pos, np.zeros_like(pos), w)
            ref = orc.Quad(pos, w).walk(pos, theta=theta, nthreads=8)
            assert np.array_equal(ctx.accel_tree(C.TREE_QUAD), ref)
            assert np.array_equal(ctx.accel_tree(C.TREE_QUAD, pos[3::5]), ref[3::5])


def test_one_pass_walk_history_survives_changes_between_steps(nb, orc, monkeypatch):
    """The wave cutting of the one-pass walk uses the previous walk's per-particle counts: they may be stale (theta, the
    application order or the arithmetic changed in between) or absent (new bodies) without changing a bit."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::4].copy(), vel[::4].copy(), w[::4].copy()
    with C.Context(0) as c:
        c.upload(pos, vel, w)
        p, v = pos, vel
        for theta, order in ((50.0, C.ORDER_AS_WRITTEN), (50.0, C.ORDER_AS_WRITTEN), (8.0, C.ORDER_AS_WRITTEN),
                             (50.0, C.ORDER_CONSISTENT), (50.0, C.ORDER_CONSISTENT)):
            c.set_params(theta=theta, order=order)
            c.update_tree(C.TREE_BVH, 0.1, 1)
            got = c.download()
            mode = orc.AS_WRITTEN if order == C.ORDER_AS_WRITTEN else orc.CONSISTENT
            p, v, w, ids, _ = orc.update_bvh(p, v, w, delta=0.1, theta=theta, mode=mode, nsteps=1, nthreads=8)
            assert np.array_equal(got[0], p) and np.array_equal(got[1], v) and np.array_equal(got[2], w)
        c.upload(pos[:5000], vel[:5000], w[:5000] * 0 + 1)         # other bodies: no history
        c.set_params(theta=50.0, order=C.ORDER_AS_WRITTEN)
        c.update_tree(C.TREE_BVH, 0.1, 2)
        p2, v2, _, _, _ = orc.update_bvh(pos[:5000], vel[:5000], np.ones(5000, np.uint32), delta=0.1, theta=50.0, mode=orc.AS_WRITTEN,
                                         nsteps=2, nthreads=8)
        got = c.download()
        assert np.array_equal(got[0], p2) and np.array_equal(got[1], v2)


def test_one_pass_walk_notices_a_history_whose_scan_wraps(nb, lab, monkeypatch, capfd):
    """A shard's slice can meet counts of older walks (another theta) whose scaled sum no longer fits 32 bits: the walk
    must notice and go on without an estimate, not cut its waves by a wrapped scan.  The test hook fills the history
    with 0xFFFFFFFF before every walk that would use it."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::3].copy(), vel[::3].copy(), w[::3].copy()
    res = []
    for poison in ("0", "1"):
        monkeypatch.setenv("NBODY_WALK_TILE_POISON", poison)
        monkeypatch.setenv("NBODY_TRACE", "1")
        with C.Context(0) as c:
            c.upload(pos, vel, w)
            c.update_tree(C.TREE_BVH, 0.1, 4)
            res.append(c.download())
        err = capfd.readouterr().err
        assert ("estimate none" in err) == (poison == "1"), err[-600:]
        assert "estimate from the last walk" in err
    for a, b in zip(*res):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ the BVH step enqueued ahead of the host (one wait per step)
@pytest.mark.parametrize("order_name", ["as_written", "consistent"])
def test_step_ahead_equals_the_plain_sequence(nb, orc, monkeypatch, capfd, order_name):
    """From the second step of a call on, a BVH step is enqueued whole — build, device-side verdict, gather, walk reading
    the node count from device memory, gated integration — with one host wait at its end (capi.hip, bvh_step_ahead).
    Same kernels on the same data: bit-identical to the phase-by-phase sequence and to the oracle."""
    C = nb._capi
    order = C.ORDER_AS_WRITTEN if order_name == "as_written" else C.ORDER_CONSISTENT
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::2].copy(), vel[::2].copy(), w[::2].copy()
    res, trees = {}, {}
    for ahead in ("1", "0"):
        monkeypatch.setenv("NBODY_STEP_AHEAD", ahead)
        monkeypatch.setenv("NBODY_TRACE", "1")
        with C.Context(0) as c:
            c.set_params(order=order)
            c.upload(pos, vel, w)
            cnt = C.Counting()
            c.update_tree(C.TREE_BVH, 0.1, 6, cnt)
            c.update_tree(C.TREE_BVH, 0.1, 1, cnt)                  # a later call starts ahead at once (the history is there)
            res[ahead] = c.download()
            assert cnt.build_bvh > 0 and cnt.sum_gravity > 0 and cnt.post_calculations > 0
            assert c.last_build_on_device() and c.tree_info().n_nodes > 0
            t = c.tree_export()                                    # the tree of the last (ahead) step is exportable
            assert t["order"].shape[0] == pos.shape[0]
            trees[ahead] = t
        err = capfd.readouterr().err
        assert (err.count("step ahead: build verdict 1") == 6) == (ahead == "1"), err[-800:]
    assert all(np.array_equal(a, b) for a, b in zip(res["1"], res["0"]))
    _bvh_export_equal(trees["1"], trees["0"])   # ... and is the tree the phase-by-phase step leaves (node count, order, geometry)
    mode = orc.AS_WRITTEN if order_name == "as_written" else orc.CONSISTENT
    rp, rv, _, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=50.0, mode=mode, nsteps=7, nthreads=16)
    assert np.array_equal(res["1"][3], rids) and np.array_equal(res["1"][0], rp) and np.array_equal(res["1"][1], rv)


def test_step_ahead_learns_how_many_levels_a_lopsided_tree_has(nb, monkeypatch, capfd):
    """A tree with more long-node levels than a balanced one's plus two (16 decades of magnitudes: the mean split peels off a
    few points at a time) fails the first speculation; the phase-by-phase build that follows records how many levels it
    took, and the steps after that are enqueued ahead with that many (+1) blind levels and stand."""
    C = nb._capi
    rng = np.random.default_rng(29)
    n = 60000
    pos = (10.0 ** rng.uniform(-8, 8, (n, 2))).astype(F32)
    monkeypatch.setenv("NBODY_TRACE", "1")
    with C.Context(0) as c:
        c.upload(pos, np.zeros_like(pos), np.ones(n, np.uint32))
        c.update_tree(C.TREE_BVH, 1e-6, 5)
        assert c.last_build_on_device()
    err = capfd.readouterr().err
    lines = [ln for ln in err.splitlines() if "step ahead: build verdict" in ln]
    assert len(lines) >= 3, err[-1500:]
    assert all("build verdict 1" in ln for ln in lines[-2:]), lines   # (the first may fail; after that the hint is right)
    levels = [int(ln.split("long-node levels")[0].split()[-1]) for ln in err.splitlines() if "long-node levels" in ln]
    assert levels and levels[0] > 7, levels   # more than a balanced tree over 60 000 points (5 levels) + 2


@pytest.mark.parametrize("hook", ["NBODY_BVH_BLIND_LEVELS", "NBODY_WALK_TILE_POISON"])
def test_step_ahead_recovers_when_its_speculation_fails(nb, lab, monkeypatch, capfd, hook):
    """Too few blind build levels (the device verdict says 'long nodes left') and a history whose scan wraps (the walk
    flags itself) both leave the step's input rows untouched — the gather writes the other set, the integration is
    gated — and the plain sequence does the step again.  Same trajectory as without the hooks."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::2].copy(), vel[::2].copy(), w[::2].copy()
    res = []
    for on in (False, True):
        if on:
            monkeypatch.setenv(hook, "1")
        monkeypatch.setenv("NBODY_TRACE", "1")
        with C.Context(0) as c:
            c.upload(pos, vel, w)
            c.update_tree(C.TREE_BVH, 0.1, 5)
            res.append(c.download())
        err = capfd.readouterr().err
        if on and hook == "NBODY_BVH_BLIND_LEVELS":
            assert "step ahead: build verdict 0" in err and "step ahead: build verdict 1" not in err
        if on and hook == "NBODY_WALK_TILE_POISON":
            assert "overflow 1" in err and "estimate none" in err
    assert all(np.array_equal(a, b) for a, b in zip(*res))


# ------------------------------------------------------------------ the f64 BVH built on the device (bvh_build64.hip)
def _bvh64_scenes(nb):
    rng = np.random.default_rng(31)
    out = {}
    for name, (pos, w) in _bvh_scenes(nb).items():
        out[name] = (pos.astype(np.float64), w)                       # f32-representable coordinates: ties at every add
    p, _, w = nb.scenes.plummer(200003, seed=91, dtype=np.float64)    # full 53-bit mantissas
    out["plummer64"] = (p, w)
    out["wide64"] = (10.0 ** rng.uniform(-8, 8, (30000, 2)), np.ones(30000, np.uint32))
    out["deep64"] = (10.0 ** rng.uniform(-100, 100, (3000, 2)), np.ones(3000, np.uint32))   # one point peeled off per level
    out["centred64"] = (rng.standard_normal((40000, 2)) * 1e4, (np.arange(40000) % 7 + 1).astype(np.uint32))
    out["tiny_n"] = (rng.random((5, 2)) * 100.0, np.ones(5, np.uint32))
    out["one"] = (np.array([[3.0, 4.0]]), np.array([9], np.uint32))
    return out


@pytest.mark.parametrize("leaf", [64, 8, 1, 200, 5000])
def test_device_bvh64_build_equals_oracle_tree(nb, orc, leaf):
    """f64 rows: the level-by-level device build (exact f64 sequential sums by the parity-map scan with 64-bit increments,
    rank-list partition, pre-order numbering from subtree sizes) gives the oracle's tree, node for node and bit for bit."""
    C = nb._capi
    with C.Context(0) as ctx:
        for name, (pos, w) in _bvh64_scenes(nb).items():
            if leaf < 8 and pos.shape[0] > 50000:
                continue
            prm = C.default_params()
            prm.leaf_size = leaf
            if C.host_tree(C.TREE_BVH, pos, w, prm)["overflow"]:
                continue   # the reference itself recurses without end here: covered by the degenerate-input tests
            ctx.set_params(theta=50.0, leaf_size=leaf)
            ctx.upload(pos, np.zeros_like(pos), w)
            ctx.accel_tree(C.TREE_BVH, pos[:4])
            # a tree deeper than the 62 levels the device follows goes to the host builder (same tree either way)
            assert ctx.last_build_on_device() == (ctx.tree_info().max_depth <= 62), (name, ctx.tree_info().max_depth)
            if name == "deep64" and leaf <= 64:
                assert ctx.tree_info().max_depth > 62
            t = ctx.tree_export()
            o = orc.BVH(pos, w, leaf_size=leaf).flat()
            for k in ("mass", "is_leaf", "first", "count", "skip"):
                assert np.array_equal(t[k], getattr(o, k)), (name, k)
            assert np.array_equal(t["geom"], o.geom, equal_nan=True), name
            assert np.array_equal(t["order"], o.ids), name
            p, _, w2, ids = ctx.download()
            assert np.array_equal(ids, o.ids) and np.array_equal(p, o.pos_perm) and np.array_equal(w2, w[o.ids]), name


def test_device_and_host_bvh64_builds_agree_over_steps(nb, lab, monkeypatch, capfd):
    """10 full f64 steps with the device build against 10 with the host build: same rows; too few blind levels (the
    build then goes on four levels at a time) changes nothing either."""
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(60000, seed=92, dtype=np.float64)
    res = []
    for host, blind in (("0", None), ("1", None), ("0", "3")):
        monkeypatch.setenv("NBODY_TREE_BUILD_HOST", host)
        if blind:
            monkeypatch.setenv("NBODY_BVH_BLIND_LEVELS", blind)
        with C.Context(0) as c:
            c.set_params(theta=0.7)
            c.upload(pos, vel, w)
            c.update_tree(C.TREE_BVH, 0.1, 10)
            assert c.last_build_on_device() == (host == "0")
            res.append(c.download())
    for other in res[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(res[0], other))


def test_device_bvh64_build_declines_what_it_cannot_express(nb, orc):
    """NaN positions (the fold's minps / maxps are order-dependent there) and more coincident points than a leaf holds
    (the reference recurses without end) go to the host builder, which handles the one and reports the other."""
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(5000, seed=93, dtype=np.float64)
    with C.Context(0) as c:
        p = pos.copy()
        p[77, 0] = np.nan
        c.upload(p, vel, w)
        c.accel_tree(C.TREE_BVH, pos[:4])
        assert not c.last_build_on_device()
        p = pos.copy()
        p[100:300] = p[100]                                            # 200 coincident points > leaf_size 64
        c.upload(p, vel, w)
        with pytest.raises(C.NBodyError) as e:
            c.accel_tree(C.TREE_BVH, pos[:4])
        assert e.value.code == C.ERR_DEGENERATE


def test_async_updates_equal_synchronous_ones(nb, orc):
    """nbody_update_tree_async_f32 returns once a step is enqueued; calls that read the rows order themselves after it.  One
    update per call, as the reference's loop makes them (main.rs:118-139), with a snapshot hand-off in between: same rows
    as the synchronous calls, same Counting shape, and nbody_wait reports the phases."""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    pos, vel, w = pos[::2].copy(), vel[::2].copy(), w[::2].copy()
    with C.Context(0) as a, C.Context(0) as b:
        for c in (a, b):
            c.upload(pos, vel, w)
        snaps = []
        for k in range(12):
            a.update_tree(C.TREE_BVH, 0.1, 1)
            b.update_tree_async(C.TREE_BVH, 0.1, 1)
            if k == 6:
                b.snapshot_begin()                       # ordered after the enqueued step, overlaps the next ones
        snaps = b.snapshot_end()
        assert snaps[4] == 7
        b.wait()
        cb = b.counting()
        assert cb.build_bvh > 0 and cb.sum_gravity > 0 and cb.post_calculations > 0
        assert all(np.array_equal(x, y) for x, y in zip(a.download(), b.download()))
        b.update_tree_async(C.TREE_QUAD, 0.1, 2)         # the plain sequence, asynchronously
        a.update_tree(C.TREE_QUAD, 0.1, 2)
        assert all(np.array_equal(x, y) for x, y in zip(a.download(), b.download()))   # download orders itself after the steps
        b.wait()
    rp, rv, _, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=50.0, mode=orc.AS_WRITTEN, nsteps=7, nthreads=16)
    assert np.array_equal(snaps[3], rids) and np.array_equal(snaps[0], rp) and np.array_equal(snaps[1], rv)


# ------------------------------------------------------------------ second half of round 3: phase clock, record fetches
def test_phase_stamps_and_event_records_time_the_same_phases(nb, lab, monkeypatch):
    """Steps enqueued ahead time their phases by the kernels' own 100 MHz clock (bvh_init / walk_scan_est_tail /
    integrate_inplace write it at the three boundaries) instead of three event records per step; NBODY_PHASE_STAMPS=0 keeps
    the events.  Same intervals: the two Counting splits agree, sum to about the wall time, and the rows are the same."""
    import time
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    out = {}
    for stamps in ("1", "0"):
        monkeypatch.setenv("NBODY_PHASE_STAMPS", stamps)
        with C.Context(0) as c:
            c.upload(pos, vel, w)
            c.update_tree(C.TREE_BVH, 0.1, 5)
            cnt = C.Counting()
            t0 = time.perf_counter()
            c.update_tree(C.TREE_BVH, 0.1, 100, cnt)
            wall = time.perf_counter() - t0
            c.update_tree(C.TREE_QUAD, 0.1, 1, cnt)            # a plain step after stamped ones: the open slot is closed first
            c.update_tree(C.TREE_BVH, 0.1, 3, cnt)             # ... and stamped steps after a plain one
            out[stamps] = (cnt.build_bvh, cnt.sum_gravity, cnt.post_calculations, wall, c.download())
    for s in out.values():
        assert s[0] > 0 and s[1] > 0 and s[2] > 0
    a, b = out["1"], out["0"]
    assert abs(a[0] - b[0]) <= 0.15 * b[0] and abs(a[1] - b[1]) <= 0.15 * b[1], (a[:3], b[:3])
    assert a[2] <= b[2] * 1.5 + 1e-3                            # (the events' gaps sat in the last phase)
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))


def test_node_record_fetch_variants_walk_the_same_walk(nb, lab, monkeypatch):
    """Node records arrive by scalar loads through the constant address space (exact walk: always; FAST: from 400 000 targets) or
    by vector loads of one address; FAST can also pin / not pin them (NBODY_WALK_FAST_REC 0 / 1 / 3).  Where they come from
    changes no bit."""
    C = nb._capi
    pos, vel, w = nb.scenes.plummer(200000, seed=61)
    for arith in (C.ARITH_AUTO, C.ARITH_FAST):
        got = []
        for srec, frec in (("1", "0"), ("0", "1"), ("1", "3")):
            monkeypatch.setenv("NBODY_WALK_SCALAR_REC", srec)
            monkeypatch.setenv("NBODY_WALK_FAST_REC", frec)
            with C.Context(0) as c:
                c.set_params(theta=50.0, arith=arith, order=C.ORDER_AS_WRITTEN)
                c.upload(pos, vel, w)
                c.update_tree(C.TREE_BVH, 0.1, 3)
                got.append(c.download())
        for other in got[1:]:
            assert all(np.array_equal(x, y) for x, y in zip(got[0], other))


def test_f64_walks_take_theta_as_the_f32_parameter_it_is(nb, orc, ctx):
    """THETA is an f32 constant upstream (main.rs:35) and `float theta` in nbody_params; an f64 walk widens that f32 value.  For
    0.5 and 50 the widening is exact; for 0.7 the double 0.7 and the widened 0.7f decide ONE borderline node of this scene's
    65 121 targets differently (row 43333: found by the extended fuzz, NBODY_FUZZ_CASES=60 NBODY_FUZZ_SEED=1).  The oracle's
    wrapper rounds theta to f32 like the clamp; with it every row agrees bit for bit, and with the double 0.7 that row does not."""
    C = nb._capi
    rng = np.random.default_rng(20261005 + 1 + 1000)        # the fuzz's stream: its second case
    n0 = int(10 ** rng.uniform(0.0, 5.3)); k0 = int(rng.integers(0, 6))
    assert (n0, k0) == (27123, 2)
    rng.random((n0, 2)); rng.integers(1, 9, n0); rng.standard_normal((n0, 2))
    n = int(10 ** rng.uniform(0.0, 5.3)); kind = int(rng.integers(0, 6))
    assert (n, kind) == (65121, 0)
    pos = (rng.random((n, 2)) * 1e5).astype(np.float64)
    w = rng.integers(1, 9, n).astype(np.uint32)
    ctx.set_params(theta=0.7, order=C.ORDER_CONSISTENT, arith=C.ARITH_AUTO)
    ctx.upload(pos, np.zeros_like(pos), w)
    acc = ctx.accel_tree(C.TREE_QUAD, pos)
    quad = orc.Quad(pos, w)
    assert np.array_equal(acc, quad.walk(pos, theta=0.7, nthreads=16))
    ctx.set_params(theta=50.0, order=C.ORDER_AS_WRITTEN)


def test_two_hundred_reference_scene_steps_equal_the_oracle(nb, orc):
    """The reference's own workload end to end (World::new's scene, 151 405 bodies, BVH, theta 50, dt 0.1, as written:
    main.rs:276-425), 200 steps in four calls on the default path — device builds, the one-pass walk cut by the previous walk's term
    counts, steps enqueued ahead of the host — against the oracle's World::update: rows, velocities, weights and the permutation bit
    for bit after every call.  (tools/soak_ref_scene.py is the same over thousands of steps: profiles/r04_soak_ref_scene_bvh.txt.)"""
    C = nb._capi
    pos, vel, w = nb.scenes.galaxy()
    o_pos, o_vel, o_w, o_ids = pos, vel, w, np.arange(pos.shape[0], dtype=np.uint32)
    with C.Context(0) as c:
        c.upload(pos, vel, w)
        for k in (1, 49, 50, 100):
            c.update_tree(C.TREE_BVH, 0.1, k)
            o_pos, o_vel, o_w, o_ids, _ = orc.update_bvh(o_pos, o_vel, o_w, nsteps=k, nthreads=16, ids=o_ids)
            p, v, w2, ids = c.download()
            assert np.array_equal(ids, o_ids) and np.array_equal(w2, o_w)
            assert np.array_equal(p.view(np.uint32), o_pos.view(np.uint32)) and np.array_equal(v.view(np.uint32), o_vel.view(np.uint32))
            assert c.last_build_on_device()
