"""Randomised differential tests: seeded adversarial inputs through the C ABI (EXACT direct, BVH and quad steps)
against the oracle, bit for bit.  Inputs mix negative and zero coordinates, duplicates, lattice points (equal
coordinates on an axis), huge and zero masses (u32 sums wrap as in release Rust), near-clamp separations."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F32 = np.float32


def _case(seed, n):
    rng = np.random.default_rng(seed)
    kind = seed % 5
    if kind == 0:      # blob around the box centre
        pos = rng.normal(50000, 8000, (n, 2))
    elif kind == 1:    # lattice with pitch 14 (the reference scene's disc): many equal x or y
        pos = np.stack([rng.integers(3000, 4000, n) * 14.0, rng.integers(3000, 4000, n) * 14.0], axis=1)
    elif kind == 2:    # both signs, straddling the origin and the root cell's edges
        pos = rng.uniform(-2e4, 1.2e5, (n, 2))
    elif kind == 3:    # two tight clusters + background: separations below the clamp radius
        pos = rng.uniform(1e4, 9e4, (n, 2))
        pos[: n // 8] = pos[0] + rng.uniform(-0.05, 0.05, (n // 8, 2))
        pos[n // 8: n // 4] = pos[n // 8] + rng.uniform(-1e-3, 1e-3, (n // 4 - n // 8, 2))
    else:              # a line (degenerate in y) with a few exact zeros
        pos = np.stack([rng.uniform(0, 1e5, n), np.full(n, 7777.0)], axis=1)
        pos[:5] = 0.0
    pos = pos.astype(F32)
    dup = rng.integers(0, n, max(1, n // 50))
    pos[dup] = pos[rng.integers(0, n, len(dup))]                 # exact duplicates (never more than a leaf holds)
    vel = rng.normal(0, 1, (n, 2)).astype(F32)
    w = rng.choice(np.array([0, 1, 1, 1, 2, 750000, 75000000, 4000000000], np.uint32), n).astype(np.uint32)
    return pos, vel, w


@pytest.mark.parametrize("seed", range(12))
def test_direct_exact_differential(nb, orc, seed):
    C = nb._capi
    n = 300 + 97 * seed
    pos, vel, w = _case(seed, n)
    with C.Context(0) as ctx:
        ctx.set_params(arith=C.ARITH_EXACT)
        ctx.upload(pos, vel, w)
        ctx.update_direct(0.1, 3)
        p, v, _, _ = ctx.download()
    rp, rv, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=3, nthreads=8)
    assert np.array_equal(p, rp, equal_nan=True) and np.array_equal(v, rv, equal_nan=True)


@pytest.mark.parametrize("walk", ["default", "one pass forced", "three passes forced"])
@pytest.mark.parametrize("order", ["as_written", "consistent"])
@pytest.mark.parametrize("seed", range(10))
def test_bvh_step_differential(nb, orc, monkeypatch, seed, order, walk):
    """`walk`: these inputs are too small for the lane = particle walks to be chosen by themselves; forcing them puts
    duplicates, wrapped and zero masses, infinite and NaN terms through the LDS tile / the term array as well."""
    if walk != "default":
        monkeypatch.setenv("NBODY_WALK_SPLIT", "3" if walk.startswith("one") else "2")
    n = 500 + 211 * seed
    pos, vel, w = _case(seed, n)
    theta = [50.0, 0.5, 2.0][seed % 3]
    leaf = [64, 16, 7][seed % 3]
    mode = orc.AS_WRITTEN if order == "as_written" else orc.CONSISTENT
    try:
        rp, rv, rw, rids, _ = orc.update_bvh(pos, vel, w, delta=0.1, theta=theta, leaf_size=leaf, mode=mode, nsteps=4, nthreads=8)
    except RuntimeError:
        world = nb.World(pos, vel, w, method="bvh", theta=theta, leaf_size=leaf, order=order)
        with pytest.raises(nb._capi.NBodyError):          # the same degenerate input must be an error on the device path too
            world.update(0.1, None, n_steps=4)
        world.close()
        return
    world = nb.World(pos, vel, w, method="bvh", theta=theta, leaf_size=leaf, order=order)
    world.update(0.1, None, n_steps=4)
    p, v, w2, ids = world.particles()
    world.close()
    assert np.array_equal(ids, rids) and np.array_equal(w2, rw)
    assert np.array_equal(p, rp, equal_nan=True) and np.array_equal(v, rv, equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("seed", range(10))
def test_quad_step_differential(nb, orc, seed, dtype):
    n = 400 + 173 * seed
    pos, vel, w = _case(seed, n)
    # duplicates of one point beyond a quad leaf's 8 slots recurse for ever upstream: keep at most 8 copies of any point
    _, first = np.unique(pos, axis=0, return_index=True)
    keep = np.sort(first)
    pos, vel, w = pos[keep].astype(dtype), vel[keep].astype(dtype), w[keep]
    theta = [0.5, 50.0, 1.0][seed % 3]
    root = (-3e4, -3e4, 2e5) if seed % 2 else (0.0, 0.0, 1e5)
    try:
        rp, rv, _ = orc.update_quad(pos, vel, w, delta=0.1, theta=theta, root=root, nsteps=4, nthreads=8)
    except RuntimeError:
        # degenerate for the reference's quad tree (unbounded recursion upstream, e.g. > 8 points beyond one corner of
        # the root cell): the device path must report it too, not hang or return something else
        world = nb.World(pos, vel, w, method="quad", theta=theta, quad_root=root)
        with pytest.raises(nb._capi.NBodyError) as e:
            world.update(0.1, None, n_steps=4)
        assert e.value.code == nb._capi.ERR_DEGENERATE
        world.close()
        return
    world = nb.World(pos, vel, w, method="quad", theta=theta, quad_root=root)
    world.update(0.1, None, n_steps=4)
    p, v, _, _ = world.particles()
    world.close()
    assert np.array_equal(p, rp, equal_nan=True) and np.array_equal(v, rv, equal_nan=True)
