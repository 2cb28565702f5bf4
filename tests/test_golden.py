"""The committed golden vectors (tests/golden/config1_1024.npz, made by tests/golden/make_golden.py from the
oracle) against (a) the oracle as it is now — guards the restatement against drift — and (b) on a GPU box, the
HIP path through the C ABI."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config1_1024.npz"))
SNAPS = (1, 10, 100)
F32 = np.float32


def test_golden_ic_is_the_seeded_plummer_set(nb):
    pos, vel, w = nb.scenes.plummer(1024, seed=0x5EED0001)
    # libm differences across hosts could move a last bit of the generator; the fixture is what the tests use
    assert np.allclose(pos, G["ic_pos"], rtol=0, atol=1e-2) and np.array_equal(w, G["ic_weight"])
    assert pos.min() > 0 and pos.max() < 100000


def test_golden_force_kats(orc):
    for r, exp in zip(G["kat_in"], G["kat_out"]):
        a = orc.pair((r[0], r[1]), (r[2], r[3]), r[4])
        assert np.array_equal(a, exp, equal_nan=True)
    assert np.array_equal(G["kat_out"][5:7], np.zeros((2, 2), F32))   # coincident, subnormal: untouched
    assert np.array_equal(G["kat_out"][8:11], np.zeros((3, 2), F32))  # inf, NaN, overflowing sum: untouched


@pytest.mark.parametrize("name,theta,mode", [("bvh_as_written_theta50", 50.0, 0), ("bvh_consistent_theta50", 50.0, 1),
                                             ("bvh_as_written_theta0p5", 0.5, 0), ("bvh_consistent_theta0p5", 0.5, 1)])
def test_oracle_reproduces_bvh_goldens(orc, name, theta, mode):
    p, v, w, ids, _ = orc.update_bvh(G["ic_pos"], G["ic_vel"], G["ic_weight"], delta=0.1, theta=theta, mode=mode, nsteps=10)
    assert np.array_equal(p, G[f"{name}_s10_pos"]) and np.array_equal(v, G[f"{name}_s10_vel"])
    assert np.array_equal(ids, G[f"{name}_s10_ids"])


def test_oracle_reproduces_quad_direct_and_tree_goldens(orc):
    p, v, _ = orc.update_quad(G["ic_pos"], G["ic_vel"], G["ic_weight"], delta=0.1, theta=0.5, nsteps=10)
    assert np.array_equal(p, G["quad_theta0p5_s10_pos"]) and np.array_equal(v, G["quad_theta0p5_s10_vel"])
    p, v, _ = orc.update_direct(G["ic_pos"], G["ic_vel"], G["ic_weight"], delta=0.1, nsteps=10)
    assert np.array_equal(p, G["direct_s10_pos"]) and np.array_equal(v, G["direct_s10_vel"])
    t = orc.BVH(G["ic_pos"], G["ic_weight"]).flat()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "ids"):
        assert np.array_equal(getattr(t, k), G[f"bvh_tree_{k}"]), k
    q = orc.Quad(G["ic_pos"], G["ic_weight"]).flat()
    for k in ("geom", "mass", "is_leaf", "depth", "child_code", "path", "first", "count", "skip", "order"):
        assert np.array_equal(getattr(q, k), G[f"quad_tree_{k}"]), k


def test_product_host_trees_equal_goldens(nb):
    C = nb._capi
    t = C.host_tree(C.TREE_BVH, G["ic_pos"], G["ic_weight"])
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip"):
        assert np.array_equal(t[k], G[f"bvh_tree_{k}"]), k
    assert np.array_equal(t["order"], G["bvh_tree_ids"])
    q = C.host_tree(C.TREE_QUAD, G["ic_pos"], G["ic_weight"])
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(q[k], G[f"quad_tree_{k}"]), k


# ------------------------------------------------------------------------------------------------ GPU vs goldens
@pytest.mark.gpu
@pytest.mark.parametrize("name,theta,order", [("bvh_as_written_theta50", 50.0, "as_written"),
                                              ("bvh_consistent_theta50", 50.0, "consistent"),
                                              ("bvh_as_written_theta0p5", 0.5, "as_written"),
                                              ("bvh_consistent_theta0p5", 0.5, "consistent")])
def test_gpu_bvh_golden_trajectories_bit_exact(nb, name, theta, order):
    """Config 1: 1 024 bodies, 100 steps of World::update on the device == the golden trajectory, bit for bit."""
    world = nb.World(G["ic_pos"], G["ic_vel"], G["ic_weight"], method="bvh", theta=theta, order=order)
    done = 0
    for s in SNAPS:
        world.update(0.1, None, n_steps=s - done)
        done = s
        p, v, _, ids = world.particles()
        assert np.array_equal(ids, G[f"{name}_s{s}_ids"]), s
        assert np.array_equal(p, G[f"{name}_s{s}_pos"]) and np.array_equal(v, G[f"{name}_s{s}_vel"]), s
    world.close()


@pytest.mark.gpu
def test_gpu_quad_golden_trajectory_bit_exact(nb):
    world = nb.World(G["ic_pos"], G["ic_vel"], G["ic_weight"], method="quad", theta=0.5)
    done = 0
    for s in SNAPS:
        world.update(0.1, None, n_steps=s - done)
        done = s
        p, v, _, _ = world.particles()
        assert np.array_equal(p, G[f"quad_theta0p5_s{s}_pos"]) and np.array_equal(v, G[f"quad_theta0p5_s{s}_vel"]), s
    world.close()


@pytest.mark.gpu
def test_gpu_direct_golden_trajectory(nb):
    """EXACT arithmetic: bit-identical to the golden; FAST arithmetic: within TRAJ_ATOL_POS after 100 steps."""
    from tests._tol import TRAJ_ATOL_POS
    world = nb.World(G["ic_pos"], G["ic_vel"], G["ic_weight"], method="direct", arith="exact")
    done = 0
    for s in SNAPS:
        world.update(0.1, None, n_steps=s - done)
        done = s
        p, v, _, _ = world.particles()
        assert np.array_equal(p, G[f"direct_s{s}_pos"]) and np.array_equal(v, G[f"direct_s{s}_vel"]), s
    world.close()
    world = nb.World(G["ic_pos"], G["ic_vel"], G["ic_weight"], method="direct", arith="fast")
    world.update(0.1, None, n_steps=100)
    p, v, _, _ = world.particles()
    world.close()
    dev = np.abs(p.astype(np.float64) - G["direct_s100_pos"]).max()
    print(f"FAST vs golden after 100 steps: max |dx| = {dev:.3e}")
    assert dev <= TRAJ_ATOL_POS


@pytest.mark.gpu
def test_gpu_accelerations_at_step0(nb):
    C = nb._capi
    with C.Context(0) as ctx:
        ctx.set_params(arith=C.ARITH_EXACT, theta=0.5)
        ctx.upload(G["ic_pos"], G["ic_vel"], G["ic_weight"])
        assert np.array_equal(ctx.accel_direct(), G["direct_acc0"])
        assert np.array_equal(ctx.accel_tree(C.TREE_QUAD), G["quad_theta0p5_acc0"])
        assert np.array_equal(ctx.accel_tree(C.TREE_BVH, G["ic_pos"]), G["bvh_theta0p5_acc0"])
        ctx.set_params(theta=50.0)
        ctx.upload(G["ic_pos"], G["ic_vel"], G["ic_weight"])
        assert np.array_equal(ctx.accel_tree(C.TREE_BVH, G["ic_pos"]), G["bvh_theta50_acc0"])


# ---------------------------------------------------------------- frames of draw() (tests/golden/frames.npz)
def _frames():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "frames.npz"))


def test_oracle_reproduces_golden_frames(nb, orc):
    f = _frames()
    pos, vel, w = nb.scenes.galaxy()
    assert np.array_equal(orc.draw(pos, vel, w, 100_000, 125), f["galaxy_125"])
    full = orc.draw(pos, vel, w, 100_000, 1250).reshape(-1, 4)
    nz = np.flatnonzero(full[:, 3])
    assert np.array_equal(nz, f["galaxy_1250_nz_index"]) and np.array_equal(full[nz], f["galaxy_1250_nz_rgba"])


@pytest.mark.gpu
def test_gpu_frames_equal_golden_frames(nb):
    C = nb._capi
    f = _frames()
    pos, vel, w = nb.scenes.galaxy()
    with C.Context(0) as ctx:
        ctx.upload(pos, vel, w)
        assert np.array_equal(ctx.render(100_000, 125), f["galaxy_125"])
        full = ctx.render(100_000, 1250).reshape(-1, 4)
        nz = np.flatnonzero(full[:, 3])
        assert np.array_equal(nz, f["galaxy_1250_nz_index"]) and np.array_equal(full[nz], f["galaxy_1250_nz_rgba"])
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "config1_1024.npz"))
    world = nb.World(g["ic_pos"], g["ic_vel"], g["ic_weight"], method="bvh", theta=50.0, order="as_written")
    try:
        world.update(0.1, None, n_steps=100)
        assert np.array_equal(world.frame(100_000, 125), f["plummer_s100_125"])
    finally:
        world.close()
