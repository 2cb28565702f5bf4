"""nbody_walk_tree_*: the force map alone (main.rs:406-416) over a CALLER'S linearised tree -- SURVEY 8b's entry point for a
host that keeps its own builder (bvh_tree.rs:56-158).  The oracle plays that host: its tree goes in, and the device walk
equals the oracle's walk bit for bit; the library's own exported tree walks back to the library's own result.  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def ctx(nb):
    c = nb._capi.Context(0)
    yield c
    c.close()


def _oracle_tree(nb, orc, kind, pos, w, leaf=64):
    C = nb._capi
    if kind == C.TREE_BVH:
        h = orc.BVH(pos, w, leaf_size=leaf)
        f = h.flat()
        order = f.ids
    else:
        h = orc.Quad(pos, w)
        f = h.flat()
        order = f.order
    return h, dict(geom=f.geom, mass=f.mass, is_leaf=f.is_leaf, first=f.first, count=f.count, skip=f.skip, order=order, kind=kind)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind_name,n,theta", [("BVH", 40, 50.0), ("BVH", 5000, 0.5), ("BVH", 30000, 50.0), ("QUAD", 5, 0.5),
                                                ("QUAD", 6000, 0.5), ("QUAD", 20000, 50.0)])
def test_the_oracles_tree_walked_on_the_device_is_the_oracles_walk(nb, orc, ctx, dtype, kind_name, n, theta):
    C = nb._capi
    kind = getattr(C, "TREE_" + kind_name)
    pos, vel, _ = nb.scenes.plummer(n, seed=61)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    w = (np.arange(n) % 6 + 1).astype(np.uint32)
    h, tree = _oracle_tree(nb, orc, kind, pos, w)
    ctx.set_params(theta=theta, leaf_size=64, arith=C.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    tg = pos[::3]
    assert np.array_equal(ctx.walk_tree(tree, tg), h.walk(tg, theta=theta, nthreads=8))
    # the rows are where a build would have left them, and the tree is the context's tree now
    p, v, w2, ids = ctx.download()
    if kind == C.TREE_BVH:
        assert np.array_equal(ids, tree["order"]) and np.array_equal(p, pos[tree["order"]]) and np.array_equal(w2, w[tree["order"]])
    else:
        assert np.array_equal(ids, np.arange(n)) and np.array_equal(p, pos)
    t = ctx.tree_export()
    for k in ("mass", "is_leaf", "first", "count", "skip"):
        assert np.array_equal(t[k], tree[k]), k
    assert np.array_equal(t["geom"], tree["geom"], equal_nan=True)
    # the particles themselves: BVH in tree order (the rows were permuted once more: by the identity), quad in row order
    tree2 = dict(tree, order=np.arange(n, dtype=np.uint32)) if kind == C.TREE_BVH else tree
    acc = ctx.walk_tree(tree2)
    assert np.array_equal(acc, h.walk(p, theta=theta, nthreads=8))


@pytest.mark.parametrize("kind_name", ["BVH", "QUAD"])
def test_the_librarys_exported_tree_walks_back_to_the_librarys_result(nb, kind_name):
    """accel_tree -> tree_export -> walk_tree on a second context holding the same upload: the same bits, at a size where the
    BVH walk is the one-pass kernel (walk_tile) and the build ran on the device."""
    C = nb._capi
    kind = getattr(C, "TREE_" + kind_name)
    pos, vel, w = nb.scenes.galaxy()                        # World::new's scene (main.rs:276-346): ~151 k bodies, three mass values
    if kind == C.TREE_QUAD:
        pos, vel, w = pos[:60000], vel[:60000], w[:60000]
    a, b = C.Context(0), C.Context(0)
    try:
        for c in (a, b):
            c.set_params(theta=50.0 if kind == C.TREE_BVH else 0.5, leaf_size=64, arith=C.ARITH_AUTO)
            c.upload(pos, vel, w)
        want = a.accel_tree(kind)
        tree = a.tree_export()
        got = b.walk_tree(tree)
        assert np.array_equal(got, want)
        pa, pb = a.download(), b.download()
        for x, y in zip(pa, pb):
            assert np.array_equal(x, y)
        # and the context goes on stepping from there as the other one does
        a.update_tree(kind, 0.1, 2)
        b.update_tree(kind, 0.1, 2)
        for x, y in zip(a.download(), b.download()):
            assert np.array_equal(x, y)
    finally:
        a.close()
        b.close()


def test_a_malformed_tree_is_refused_before_it_reaches_the_device(nb, orc, ctx):
    C = nb._capi
    n = 5000
    pos, vel, _ = nb.scenes.plummer(n, seed=62)
    w = np.ones(n, np.uint32)
    h, tree = _oracle_tree(nb, orc, C.TREE_BVH, pos, w)
    ctx.set_params(theta=0.5, leaf_size=64, arith=C.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    inner = np.flatnonzero(tree["is_leaf"] == 0)
    for key, idx, value in (("skip", inner[3], inner[3] - 1), ("first", 5, -4), ("order", 0, n + 7), ("count", inner[2], 1)):
        bad = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in tree.items()}
        bad[key][idx] = value
        with pytest.raises(C.NBodyError) as e:
            ctx.walk_tree(bad, pos[:10])
        assert e.value.code == C.ERR_INVALID and "walk_tree" in str(e.value)
    _, fewer = _oracle_tree(nb, orc, C.TREE_BVH, pos[:-1], w[:-1])  # a tree over fewer rows than the context holds
    fewer["order"] = np.concatenate([fewer["order"], np.array([n - 1], np.uint32)])
    with pytest.raises(C.NBodyError) as e:
        ctx.walk_tree(fewer, pos[:10])
    assert "every particle" in str(e.value)
    # nothing was touched: the rows are the upload, and the good tree still walks
    p, _, _, ids = ctx.download()
    assert np.array_equal(ids, np.arange(n)) and np.array_equal(p, pos)
    assert np.array_equal(ctx.walk_tree(tree, pos[:100]), h.walk(pos[:100], theta=0.5, nthreads=8))


def test_fast_arithmetic_over_a_callers_tree_keeps_the_tolerance(nb, orc, ctx):
    from tests._tol import check_fast
    C = nb._capi
    n = 40000
    pos, vel, _ = nb.scenes.plummer(n, seed=63)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    h, tree = _oracle_tree(nb, orc, C.TREE_BVH, pos, w)
    ctx.set_params(theta=50.0, leaf_size=64, arith=C.ARITH_FAST)
    ctx.upload(pos, vel, w)
    acc = ctx.walk_tree(tree)
    f = h.flat()
    ref64, norm = h.walk_ref(f.pos_perm, theta=50.0, nthreads=8)
    check_fast(acc, ref64, np.maximum(norm, 1e-300), label=" caller's tree, FAST")
    ctx.set_params(arith=C.ARITH_AUTO)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_a_kept_quad_tree_is_walked_as_the_host_keeps_it(nb, orc, ctx, dtype):
    """QuadTree::empty / prune (quad_tree.rs:66-137) keep a tree from step to step, and its cells are then not a fresh build's
    (a root stays a root however few points it holds).  The library rebuilds every step and has no such tree of its own: a
    host that keeps one hands it to nbody_walk_tree_* and gets ITS tree's interaction lists."""
    C = nb._capi
    n = 20000
    pos, vel, _ = nb.scenes.plummer(n, seed=64)
    pos, vel = pos.astype(dtype), vel.astype(dtype)
    w = (np.arange(n) % 3 + 1).astype(np.uint32)
    q = orc.Quad(pos, w)
    fresh_nodes = len(q.flat().skip)
    rng = np.random.default_rng(23)
    moved = (pos * 0.5 + 25000.0 + rng.standard_normal(pos.shape) * 50.0).astype(dtype)   # the cloud shrinks: its old cells thin out
    q.reuse(moved, w)
    f = q.flat()
    assert len(f.skip) != len(orc.Quad(moved, w).flat().skip) and fresh_nodes > 0
    tree = dict(geom=f.geom, mass=f.mass, is_leaf=f.is_leaf, first=f.first, count=f.count, skip=f.skip, order=f.order, kind=C.TREE_QUAD)
    ctx.set_params(theta=0.5, arith=C.ARITH_AUTO)
    ctx.upload(moved, vel, w)
    assert np.array_equal(ctx.walk_tree(tree), q.walk(moved, theta=0.5, nthreads=8))
    assert not np.array_equal(ctx.accel_tree(C.TREE_QUAD), q.walk(moved, theta=0.5, nthreads=8))   # the library's own (fresh) tree differs


@pytest.mark.parametrize("n", [1, 2, 64, 65])
def test_the_smallest_trees_walk_too(nb, orc, ctx, n):
    """BVHTree::from always makes a Root (main.rs:400 calls it unconditionally), so even one body comes as a root with two
    leaves, one of them empty (offset MAX, size -MAX, NaN centre of gravity: bvh_tree.rs:40-54)."""
    C = nb._capi
    pos = np.array([[10.0 + 3 * k, 20.0 + 7 * (k % 5)] for k in range(n)], F32)
    vel = np.zeros_like(pos)
    w = np.arange(1, n + 1, dtype=np.uint32)
    h, tree = _oracle_tree(nb, orc, C.TREE_BVH, pos, w)
    ok, why = C.tree_validate(tree, n)
    assert ok, why
    ctx.set_params(theta=0.5, leaf_size=64, arith=C.ARITH_AUTO)
    ctx.upload(pos, vel, w)
    tg = np.array([[0.0, 0.0], [11.0, 21.0], [1e4, -3.0]], F32)
    assert np.array_equal(ctx.walk_tree(tree, tg), h.walk(tg, theta=0.5))


def test_a_callers_tree_behind_a_sharded_handle(nb, orc):
    """The same call on a context that fronts several devices (here: the one device three times, peer copies): the first
    replica walks, the others are brought up to its rows, and the steps that follow equal the plain context's."""
    C = nb._capi
    n = 20000
    pos, vel, _ = nb.scenes.plummer(n, seed=65)
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    h, tree = _oracle_tree(nb, orc, C.TREE_BVH, pos, w)
    want = h.walk(pos[::5], theta=0.5, nthreads=8)
    prm = dict(theta=0.5, leaf_size=64, arith=C.ARITH_AUTO)
    with C.Context(0) as s, C.MultiContext([0, 0, 0], C.EXCHANGE_PEER, 2) as m:
        for c in (s, m):
            c.set_params(**prm)
            c.upload(pos, vel, w)
            assert np.array_equal(c.walk_tree(tree, pos[::5]), want)
            c.update_tree(C.TREE_BVH, 0.1, 2)
            c.update_direct(0.1, 1)
        for x, y in zip(s.download(), m.download()):
            assert np.array_equal(x, y)
