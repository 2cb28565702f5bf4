"""Known-answer and invariant tests that pin the CPU oracle.  No GPU.

The reference has no tests or fixtures ("parity unpinned"), so the oracle is pinned by (a) hand-derived exact
rationals for the force law, (b) an independently written numpy restatement, (c) structural invariants of the
trees that follow from the reference source regardless of third-party details.
"""
from fractions import Fraction

import numpy as np
import pytest

from tests import _np_restatement as npr

F32 = np.float32


def f32(fr):
    return F32(float(fr))


# ---------------------------------------------------------------- a1: calculate_gravity  (main.rs:234-253)
KATS = [
    # p1, p2, mass, expected (exact rationals before the final f32 rounding)
    ((0, 0), (3, 4), 2, (Fraction(6, 175), Fraction(8, 175))),
    ((0, 0), (1, 0), 1, (Fraction(1), Fraction(0))),
    ((0, 0), (-3, -4), 1, (Fraction(-3, 175), Fraction(-4, 175))),
    ((10, 20), (7, 24), 750000, (Fraction(-90000, 7), Fraction(120000, 7))),
]


@pytest.mark.parametrize("p1,p2,m,exp", KATS)
def test_force_kat_f32(orc, p1, p2, m, exp):
    a = orc.pair(p1, p2, m)
    # the reference rounds three times (mul, mul, div); all operands here are small integers, so only the
    # division rounds: the result is the correctly rounded quotient
    assert a[0] == f32(exp[0]) and a[1] == f32(exp[1])


@pytest.mark.parametrize("p1,p2,m,exp", KATS)
def test_force_kat_f64(orc, p1, p2, m, exp):
    a = orc.pair(p1, p2, m, dtype=np.float64)
    assert a[0] == float(exp[0]) and a[1] == float(exp[1])


def test_force_clamp(orc):
    # d^2 = 1e-4 < 0.001 -> clamped: a = dx*m / (|dx| * 0.001) = 1000 (up to f32 rounding of 0.01 and 0.001)
    a = orc.pair((0, 0), (0.01, 0), 1)
    expect = (F32(0.01) * F32(1)) / (F32(0.01) * F32(0.001))
    assert a[0] == expect and a[1] == 0
    assert abs(float(a[0]) - 1000.0) < 1e-3


def test_force_accumulates(orc):
    a = orc.pair((0, 0), (3, 4), 2, acc=(1.0, -1.0))
    assert a[0] == F32(1.0) + f32(Fraction(6, 175)) and a[1] == F32(-1.0) + f32(Fraction(8, 175))


@pytest.mark.parametrize("p2", [(0.0, 0.0), (1e-39, 0.0), (0.0, -1e-40), (np.inf, 0.0), (np.nan, 1.0),
                                (3e38, 3e38)])
def test_force_skips_non_normal_sum(orc, p2):
    # identical points (sum 0), subnormal sum, inf, NaN, and |dx|+|dy| overflowing to inf: accumulator untouched
    a = orc.pair((0.0, 0.0), p2, 5, acc=(0.25, -0.5))
    assert a[0] == F32(0.25) and a[1] == F32(-0.5)


def test_force_smallest_normal_sum_is_not_skipped(orc):
    tiny = float(np.finfo(np.float32).tiny)
    a = orc.pair((0.0, 0.0), (tiny, 0.0), 1)
    # d2 underflows to 0 -> clamp; den = tiny*0.001 is subnormal but the division is still carried out
    expect = (F32(tiny) * F32(1)) / (F32(tiny) * F32(0.001))
    assert a[0] == expect and a[0] > 900


def test_force_matches_numpy_restatement(orc):
    rng = np.random.default_rng(1)
    src = (rng.random((4096, 2)) * 1e5).astype(F32)
    src[7] = src[0]                      # coincident with the target
    src[9] = src[0] + F32(0.01)          # inside the clamp
    w = rng.integers(1, 1000, 4096).astype(np.uint32)
    terms = npr.pair_terms(src[0], src, w)
    for j in range(0, 4096, 37):
        a = orc.pair(src[0], src[j], w[j])
        assert a[0] == terms[j, 0] and a[1] == terms[j, 1], j
    for j in (0, 7, 9):
        a = orc.pair(src[0], src[j], w[j])
        assert a[0] == terms[j, 0] and a[1] == terms[j, 1]


def test_direct_sum_matches_numpy_restatement(orc):
    rng = np.random.default_rng(2)
    pos = (rng.random((300, 2)) * 1e3).astype(F32)
    w = rng.integers(1, 50, 300).astype(np.uint32)
    tg = [0, 1, 17, 299]
    acc, _ = orc.direct_accel(pos, w, targets=tg)
    ref = npr.direct_accel_seq(pos, w, tg)
    assert np.array_equal(acc.astype(F32), ref)


def test_direct_f64_accumulation_close_to_native(orc, nb):
    pos, vel, w = nb.scenes.plummer(2048, seed=11)
    a32, _ = orc.direct_accel(pos, w, targets=np.arange(64))
    a64, norm = orc.direct_accel(pos, w, targets=np.arange(64), accum="f64")
    err = np.abs(a32 - a64).sum(axis=1)
    assert np.all(err <= 2e-5 * norm)


# ---------------------------------------------------------------- a3(4): integrate  (main.rs:419-423)
def test_update_direct_is_semi_implicit_euler(orc):
    pos = np.array([[0, 0], [3, 4]], F32)
    vel = np.array([[1, 2], [0, 0]], F32)
    w = np.array([2, 2], np.uint32)
    dt = F32(0.1)
    p1, v1, _ = orc.update_direct(pos, vel, w, delta=0.1, nsteps=1)
    a0 = np.array([f32(Fraction(6, 175)), f32(Fraction(8, 175))])
    v0 = vel[0] + a0 * dt
    assert np.array_equal(v1[0], v0)
    assert np.array_equal(p1[0], pos[0] + v0 * dt)
    assert np.array_equal(v1[1], -(a0 * dt))     # equal and opposite (equal masses)


# ---------------------------------------------------------------- a4-a6: BVH  (bvh_tree.rs)
def _bvh_case(nb, n, seed, leaf=64):
    pos, vel, w = nb.scenes.plummer(n, seed=seed)
    w = (np.arange(n) % 7 + 1).astype(np.uint32)
    return pos, vel, w


@pytest.mark.parametrize("n,leaf", [(1024, 64), (5000, 64), (777, 16), (40, 64), (65, 64)])
def test_bvh_invariants(orc, nb, n, leaf):
    pos, _, w = _bvh_case(nb, n, 3)
    t = orc.BVH(pos, w, leaf_size=leaf).flat()
    assert not t.overflow
    m = len(t.mass)
    # permutation is a bijection and positions moved with their ids
    assert np.array_equal(np.sort(t.ids), np.arange(n))
    assert np.array_equal(t.pos_perm, pos[t.ids])
    wp = w[t.ids]
    # the top call is unconditional: node 0 is a Root even for n <= leaf (main.rs:400)
    assert t.is_leaf[0] == 0
    assert t.mass[0] == w.sum()
    leaves = np.flatnonzero(t.is_leaf)
    assert np.all(t.count[leaves] <= leaf)
    assert t.count[leaves].sum() == n
    # leaves tile the permuted array in pre-order
    assert np.array_equal(t.first[leaves], np.concatenate([[0], np.cumsum(t.count[leaves])[:-1]]))
    MAXV = np.finfo(F32).max
    for i in range(m):
        if t.is_leaf[i]:
            sl = slice(t.first[i], t.first[i] + t.count[i])
            p = t.pos_perm[sl]
            if len(p):
                mn = np.minimum(MAXV, p.min(axis=0))
                mx = np.maximum(F32(0), p.max(axis=0))
                assert np.array_equal(t.geom[i, 0:2], mn) and np.array_equal(t.geom[i, 2:4], mx - mn)
                # leaf COG = unweighted sequential mean (bvh_tree.rs:104-107)
                acc = np.zeros(2, F32)
                for q in p:
                    acc = acc + q
                assert np.array_equal(t.geom[i, 4:6], acc / F32(len(p)))
            assert t.mass[i] == wp[sl].sum()
            assert t.skip[i] == i + 1
        else:
            l, r = i + 1, t.skip[i + 1]
            assert t.skip[i] == t.skip[r]
            assert t.mass[i] == t.mass[l] + t.mass[r]
            big = t.geom[l, 4:6] * F32(t.mass[l]) + t.geom[r, 4:6] * F32(t.mass[r])
            assert np.array_equal(t.geom[i, 4:6], big / F32(t.mass[i]))


def _subtree_particles(t, i):
    """[first, end) of the permuted array covered by node i."""
    j = i
    while not t.is_leaf[j]:
        j += 1
    first = t.first[j]
    k = t.skip[i] - 1                     # last node of the subtree is a leaf
    return first, t.first[k] + t.count[k]


def test_bvh_split_predicate(orc, nb):
    """Every split obeys its predicate against the parent's recomputed mean; left = the 'greater' side;
    axis choice follows bvh_tree.rs:70-77."""
    pos, _, w = _bvh_case(nb, 3000, 5)
    t = orc.BVH(pos, w).flat()
    for i in np.flatnonzero(t.is_leaf == 0):
        a, b = _subtree_particles(t, i)
        l, r = i + 1, t.skip[i + 1]
        la, lb = _subtree_particles(t, l)
        ra, rb = _subtree_particles(t, r)
        assert (la, rb) == (a, b) and lb == ra
        left, right = t.pos_perm[la:lb], t.pos_perm[ra:rb]
        allp = t.pos_perm[a:b]
        lx = np.all(left[:, 0] > right[:, 0].max()) if len(left) and len(right) else True
        ly = np.all(left[:, 1] > right[:, 1].max()) if len(left) and len(right) else True
        assert lx or ly
        # box = fold(min from MAX, max from 0)
        assert np.array_equal(t.geom[i, 0:2], allp.min(axis=0))
        assert np.array_equal(t.geom[i, 2:4], np.maximum(F32(0), allp.max(axis=0)) - allp.min(axis=0))


def test_bvh_split_axis_rule_small(orc):
    """Hand-checkable 130-point case: x is well balanced around its mean, y is not -> vert > hori -> split on x."""
    n = 130
    x = np.arange(n, dtype=F32)
    y = np.where(np.arange(n) < 120, F32(1.0), F32(1000.0)).astype(F32)
    pos = np.stack([x, y], axis=1)
    t = orc.BVH(pos, None).flat()
    mean_x = F32(0)
    for v in x:
        mean_x = F32(mean_x + v)
    mean_x = mean_x / F32(n)
    la, lb = _subtree_particles(t, 1)
    assert np.all(t.pos_perm[la:lb, 0] > mean_x)
    assert lb - la == int((x > mean_x).sum())


def test_bvh_max_fold_starts_at_zero(orc):
    """bvh_tree.rs:42,59: max starts at 0.0, so all-negative coordinates give size = 0 - min."""
    pos = np.array([[-5, -7], [-1, -2], [-3, -9]], F32)
    t = orc.BVH(pos, None).flat()
    assert np.array_equal(t.geom[0, 0:2], [-5, -9])
    assert np.array_equal(t.geom[0, 2:4], [5, 9])


def test_bvh_hoare_partition_order(orc):
    """Order inside a side after the two-pointer partition (crate partition 0.1.2, restated): the k-th
    misplaced element from the left swaps with the k-th misplaced from the right."""
    n = 70   # > 64 so the root's sides are leaves whose slice order is visible
    x = np.arange(n, dtype=F32)
    rng = np.random.default_rng(0)
    rng.shuffle(x)
    pos = np.stack([x, np.zeros(n, F32)], axis=1)   # y all equal: cy = 0 -> vert = 35 > hori -> split on x
    t = orc.BVH(pos, None).flat()
    s = F32(0)
    for v in x:
        s = F32(s + v)
    mean = s / F32(n)
    pred = x > mean
    arr = list(range(n))
    l, r = 0, n - 1
    while True:
        while l < n and pred[arr[l]]:
            l += 1
        while r > 0 and not pred[arr[r]]:
            r -= 1
        if l >= r:
            break
        arr[l], arr[r] = arr[r], arr[l]
    assert np.array_equal(t.ids, np.array(arr, np.uint32))


def test_bvh_degenerate_hits_depth_cap(orc):
    pos = np.tile(np.array([[5.0, 5.0]], F32), (100, 1))
    t = orc.BVH(pos, None).flat()
    assert t.overflow


# ---------------------------------------------------------------- a2: walker  (main.rs:348-386)
def test_walk_theta0_equals_direct_sum_multiset(orc, nb):
    """theta = 0: s^2 < d^2*0 never holds -> every leaf is visited -> same multiset of terms as the direct sum."""
    pos, _, w = _bvh_case(nb, 2000, 9)
    bvh = orc.BVH(pos, w)
    tg = pos[:50]
    acc, st = bvh.walk(tg, theta=0.0, stats=True)
    assert st[1] == 0 and st[2] == 50 * 2000
    ref, norm = orc.direct_accel(pos, w, target_pos=tg, accum="f64")
    assert np.all(np.abs(acc - ref).sum(axis=1) <= 2e-5 * norm)


def test_walk_theta0_bit_exact_when_order_is_the_same(orc, nb):
    """On the permuted array the leaf order IS ascending j, so theta=0 walk == sequential direct sum, bit for bit."""
    pos, _, w = _bvh_case(nb, 1500, 10)
    bvh = orc.BVH(pos, w)
    t = bvh.flat()
    tg = t.pos_perm[:40]
    acc = bvh.walk(tg, theta=0.0)
    ref, _ = orc.direct_accel(t.pos_perm, w[t.ids], target_pos=tg)
    assert np.array_equal(acc, ref.astype(F32))


def test_walk_accepts_far_nodes(orc, nb):
    pos, _, w = _bvh_case(nb, 4000, 12)
    bvh = orc.BVH(pos, w)
    far = np.array([[9e5, 9e5]], F32)
    acc, st = bvh.walk(far, theta=0.5, stats=True)
    assert st[0] == 1 and st[1] == 1 and st[2] == 0     # root accepted at once
    t = bvh.flat()
    ref = orc.pair(far[0], t.geom[0, 4:6], F32(t.mass[0]))
    assert np.array_equal(acc[0], ref)


def test_walk_theta_monotone_work(orc, nb):
    pos, _, w = _bvh_case(nb, 4000, 13)
    bvh = orc.BVH(pos, w)
    visits = [int(bvh.walk(pos[:200], theta=th, stats=True)[1][0]) for th in (0.0, 0.5, 2.0, 50.0)]
    assert visits == sorted(visits, reverse=True)


# ---------------------------------------------------------------- World::update modes  (main.rs:388-425)
def test_update_bvh_modes(orc, nb):
    pos, vel, w = nb.scenes.plummer(1024, seed=21)
    pa, va, wa, ia, _ = orc.update_bvh(pos, vel, w, mode=orc.AS_WRITTEN, nsteps=1)
    pc, vc, wc, ic, _ = orc.update_bvh(pos, vel, w, mode=orc.CONSISTENT, nsteps=1)
    assert np.array_equal(ia, ic)                         # same permutation
    assert np.array_equal(np.sort(ia), np.arange(1024))
    # consistent: row k is particle ids[k] advanced by ITS OWN acceleration
    bvh = orc.BVH(pos, w)
    t = bvh.flat()
    acc = bvh.walk(t.pos_perm, theta=50.0)
    dt = F32(0.1)
    v = vel[t.ids] + acc * dt
    assert np.array_equal(vc, v) and np.array_equal(pc, t.pos_perm + v * dt)
    # as written: row k gets the acceleration computed for the snapshot's row k (main.rs:406-423)
    acc_snap = bvh.walk(pos, theta=50.0)
    v = vel[t.ids] + acc_snap * dt
    assert np.array_equal(va, v) and np.array_equal(pa, t.pos_perm + v * dt)
    assert not np.array_equal(va, vc)


# ---------------------------------------------------------------- a8: quad tree  (quad_tree.rs)
def test_quad_invariants(orc, nb):
    n = 3000
    pos, _, _ = nb.scenes.plummer(n, seed=4)
    w = (np.arange(n) % 5 + 1).astype(np.uint32)
    q = orc.Quad(pos, w).flat()
    assert not q.overflow
    assert np.array_equal(np.sort(q.order), np.arange(n))
    leaves = np.flatnonzero(q.is_leaf)
    assert np.all(q.count[leaves] <= 8) and np.all(q.count[leaves] >= 1)
    assert q.mass[0] == w.sum()
    for i in range(len(q.mass)):
        ids = q.order[q.first[i]:q.first[i] + q.count[i]]
        ox, oy, h = q.geom[i, 0:3]
        if q.is_leaf[i]:
            assert np.all(np.diff(ids.astype(np.int64)) > 0)      # insertion (index) order inside a leaf
            acc = np.zeros(2, F32)
            for k in ids:
                acc = acc + pos[k]
            assert np.array_equal(q.geom[i, 3:5], acc / F32(len(ids)))
            assert q.mass[i] == w[ids].sum()
        else:
            assert q.count[i] > 8                                  # a cell is internal iff it ever held > 8
            half = h / F32(2.0)
            xm, ym = ox + half, oy + half
            c = i + 1
            big = np.zeros(2, F32)
            msum = 0
            while c < q.skip[i]:
                code = q.child_code[c]
                cids = q.order[q.first[c]:q.first[c] + q.count[c]]
                assert np.all((2 * (pos[cids, 1] > ym) + (pos[cids, 0] > xm)) == code)   # quad_tree.rs:176-179
                exp_off = (ox + (half if code & 1 else F32(0)), oy + (half if code & 2 else F32(0)))
                assert q.geom[c, 0] == exp_off[0] and q.geom[c, 1] == exp_off[1] and q.geom[c, 2] == half
                assert q.depth[c] == q.depth[i] + 1
                big = big + q.geom[c, 3:5] * F32(q.mass[c])
                msum += int(q.mass[c])
                c = q.skip[c]
            assert q.mass[i] == msum
            assert np.array_equal(q.geom[i, 3:5], big / F32(msum))


def test_quad_cells_independent_of_insertion_order(orc, nb):
    n = 2000
    pos, _, _ = nb.scenes.plummer(n, seed=6)
    a = orc.Quad(pos).flat()
    perm = np.random.default_rng(0).permutation(n)
    b = orc.Quad(pos[perm]).flat()
    ka = sorted(zip(a.depth.tolist(), a.path.tolist(), a.is_leaf.tolist(), a.count.tolist()))
    kb = sorted(zip(b.depth.tolist(), b.path.tolist(), b.is_leaf.tolist(), b.count.tolist()))
    assert ka == kb


def test_quad_walk_theta0_is_direct(orc, nb):
    n = 1500
    pos, _, w = nb.scenes.plummer(n, seed=8)
    q = orc.Quad(pos, w)
    acc, st = q.walk(pos[:30], theta=0.0, stats=True)
    assert st[1] == 0 and st[2] == 30 * n
    ref, norm = orc.direct_accel(pos, w, target_pos=pos[:30], accum="f64")
    assert np.all(np.abs(acc - ref).sum(axis=1) <= 2e-5 * norm)


def test_f64_restatement_agrees_with_f32_loosely(orc, nb):
    pos, _, w = nb.scenes.plummer(1024, seed=14)
    a32 = orc.BVH(pos, w).walk(pos[:64], theta=0.5)
    a64 = orc.BVH(pos.astype(np.float64), w).walk(pos[:64].astype(np.float64), theta=0.5)
    # different rounding can flip a borderline split, so this is only a sanity bound
    assert np.median(np.abs(a32 - a64) / (np.abs(a64) + 1e-12)) < 1e-3


# ---------------------------------------------------------------- second reading of trees + walker (plain Python)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,leaf,theta", [(300, 64, 0.5), (700, 16, 50.0), (1200, 64, 2.0)])
def test_oracle_bvh_agrees_with_python_restatement(orc, nb, dtype, n, leaf, theta):
    """The C++ oracle and an independently written Python reading of bvh_tree.rs + main.rs:348-386 must agree bit for
    bit: permutation of the particles, and the walked acceleration of every sampled target."""
    pos, _, _ = nb.scenes.plummer(n, seed=101, dtype=dtype)
    w = (np.arange(n) % 9 + 1).astype(np.uint32)
    w[3] = 75_000_000
    py = npr.PyBVH(pos, w, leaf_size=leaf, dtype=dtype)
    bvh = orc.BVH(pos, w, leaf_size=leaf)
    assert np.array_equal(py.ids(), bvh.flat().ids)
    tg = pos[::17]
    ref = bvh.walk(tg, theta=theta)
    got = np.array([py.walk(p, theta) for p in tg], dtype=dtype)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,theta", [(200, 0.5), (900, 0.5), (900, 50.0)])
def test_oracle_quad_agrees_with_python_restatement(orc, nb, dtype, n, theta):
    pos, _, _ = nb.scenes.plummer(n, seed=102, dtype=dtype)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    py = npr.PyQuad(pos, w, dtype=dtype)
    q = orc.Quad(pos, w)
    tg = pos[::13]
    ref = q.walk(tg, theta=theta)
    got = np.array([py.walk(p, theta) for p in tg], dtype=dtype)
    assert np.array_equal(got, ref)
    assert int(py.root["mass"] if not py.root["leaf"] else 0) == int(q.flat().mass[0])


# ---------------------------------------------------------------- draw()  (main.rs:41-72)
def test_draw_known_answers(orc):
    """Hand-derived pixels: cell = 100000 / 1250 = 80 world units."""
    pos = np.array([[85, 170], [85.5, 171], [99999, 0], [-1, 5], [100000, 5], [500, 500], [500, 501], [np.nan, 3]], F32)
    vel = np.array([[0.3, -0.4], [1, 1], [100, 0], [0, 0], [0, 0], [0, 0], [2, 2], [0, 0]], F32)
    w = np.array([1, 1, 1, 1, 1, 75_000_000, 1, 1], np.uint32)
    f = orc.draw(pos, vel, w)
    assert f.shape == (1250, 1250, 4)
    # rows 0 and 1 share pixel (x 1, y 2): the later row colours it, (|1|+|1|)*10 = 20 -> v = 0x10 + 20; alpha 2 * 10
    assert tuple(f[2, 1]) == (255, 255 - 36, 255 - 36, 20)
    # (|100|)*10 saturates `as u8`, then .min(0xef): v = 0xff
    assert tuple(f[0, 1249]) == (255, 0, 0, 10)
    # weight > 10 paints green and a later light row leaves it alone; out-of-bounds and NaN rows paint nothing
    assert tuple(f[6, 6]) == (0, 255, 0, 255)
    assert np.count_nonzero(f[..., 3]) == 3


def test_draw_alpha_saturates_at_250_and_heavy_overrides(orc):
    n = 40
    pos = np.tile(np.array([[1000.5, 2000.5]], F32), (n, 1))
    vel = np.zeros((n, 2), F32)
    vel[-1] = (0.5, 0.25)
    w = np.ones(n, np.uint32)
    f = orc.draw(pos, vel, w)
    assert tuple(f[25, 12]) == (255, 255 - (0x10 + 7), 255 - (0x10 + 7), 250)   # 25 increments of 10, then stuck
    w[3] = 11                                                                     # one heavy row anywhere
    assert tuple(orc.draw(pos, vel, w)[25, 12]) == (0, 255, 0, 255)
    w[3] = 10                                                                     # "> 10" is strict
    assert tuple(orc.draw(pos, vel, w)[25, 12])[3] == 250


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_draw_matches_independent_python_loop(orc, dtype):
    rng = np.random.default_rng(12)
    n = 4000
    pos = (rng.random((n, 2)) * 1.2e5 - 1e4).astype(dtype)       # some outside the box
    pos[:300] = (rng.integers(0, 40, (300, 2)) * 80.0 + 3).astype(dtype)   # crowded pixels
    vel = (rng.standard_normal((n, 2)) * 3).astype(dtype)
    vel[5] = (np.nan, 1)
    vel[6] = (1e30, 0)
    w = np.where(rng.random(n) < 0.02, 750_000, 1).astype(np.uint32)
    got = orc.draw(pos, vel, w, 100_000, 1250)
    want = npr.draw(pos, vel, w, 100_000, 1250)
    assert np.array_equal(got, want)
    assert np.array_equal(orc.draw(pos, vel, w, 100_000, 100), npr.draw(pos, vel, w, 100_000, 100))


# ---------------------------------------------------------------- QuadTree::empty / prune  (quad_tree.rs:66-137)
def _quad_cells(flat):
    """The oracle's flat quad tree as PyQuad.cells() lists a tree: (path of child codes, is a leaf, ids of its points)."""
    out = []
    for i in range(len(flat.skip)):
        d = int(flat.depth[i])
        path = tuple((int(flat.path[i]) >> (2 * (d - 1 - k))) & 3 for k in range(d))
        ids = tuple(int(x) for x in flat.order[flat.first[i]:flat.first[i] + flat.count[i]]) if flat.is_leaf[i] else ()
        out.append((path, bool(flat.is_leaf[i]), ids))
    return out


def test_quad_empty_and_prune_known_answers(orc):
    """Root cell (0, 0, 100).  Eight points in the lower-left quarter and one in the upper-right: the ninth insert subdivides
    the root into two leaves (child codes 0 and 3)."""
    ll = np.array([[5 + 5 * k, 10 + 3 * k] for k in range(8)], F32)          # x, y < 50: child 0
    pos = np.vstack([ll, [[80, 90]]]).astype(F32)                            # child 3
    w = np.ones(9, np.uint32)
    q = orc.Quad(pos, w, root=(0.0, 0.0, 100.0))
    f = q.flat()
    assert list(f.is_leaf) == [0, 1, 1] and list(f.child_code) == [0, 0, 3] and list(f.count) == [9, 8, 1]
    assert q.empty() == 3                  # the root and its two leaves (:66-89)
    assert q.empty() == 1                  # a root without mass is not entered (:77-79)
    # the same points again: the kept cells take them back, nothing to prune
    assert q.reuse(pos, w) == (1, 0)       # (empty() found the root still without mass: one cell)
    assert _quad_cells(q.flat()) == _quad_cells(f)
    # the ninth point joins the others: child 0 is full and subdivides, child 3 is an empty leaf and is pruned
    moved = pos.copy()
    moved[8] = (12.0, 40.0)
    emptied, pruned = q.reuse(moved, w)
    assert emptied == 3 and pruned == 1
    g = q.flat()
    assert g.is_leaf[0] == 0 and g.is_leaf[1] == 0 and g.mass[0] == 9 and g.mass[1] == 9
    assert 3 not in list(g.child_code[g.depth == 1])           # the flag bit of the pruned child is flipped: no cell 3 any more
    # everything moves to the upper right: child 0, a ROOT by now, ends without mass and is dropped as one child (:118-124)
    ur = (pos * F32(0.2) + F32(70.0)).astype(F32)
    emptied, pruned = q.reuse(ur, w)
    assert pruned == 1 and emptied == len(g.skip)
    k = q.flat()
    assert list(k.child_code[k.depth == 1]) == [3] and k.mass[0] == 9
    # a fresh build over the same points has fewer cells where the kept tree stays subdivided, never more points per leaf
    assert len(orc.Quad(ur, w, root=(0.0, 0.0, 100.0)).flat().skip) <= len(k.skip)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_kept_quad_tree_agrees_with_python_restatement(orc, nb, dtype):
    """Three steps of a tree that is kept (empty, insert, calculate_gravity, prune): the counts the two calls return, every
    cell with its points in order, and the walk, against the independent Python reading of quad_tree.rs:66-137."""
    n = 700
    pos, _, _ = nb.scenes.plummer(n, seed=104, dtype=dtype)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    py = npr.PyQuad(pos, w, dtype=dtype)
    q = orc.Quad(pos, w)
    rng = np.random.default_rng(17)
    for step in range(3):
        pos = (pos + rng.standard_normal(pos.shape) * (300.0 * (step + 1))).astype(dtype)
        if step == 2:
            pos[: n // 2] = (pos[: n // 2] * 0.25 + 60000.0).astype(dtype)      # half of the points leave their cells for one corner
        assert q.reuse(pos, w) == py.reuse(pos, w)
        assert _quad_cells(q.flat()) == py.cells()
        tg = pos[::29]
        got = np.array([py.walk(p, 0.5) for p in tg], dtype=dtype)
        assert np.array_equal(got, q.walk(tg, theta=0.5))


def test_kept_quad_tree_over_unmoved_points_is_the_fresh_build(orc, nb):
    n = 3000
    pos, _, _ = nb.scenes.plummer(n, seed=105)
    w = np.ones(n, np.uint32)
    q = orc.Quad(pos, w)
    f = q.flat()
    emptied, pruned = q.reuse(pos, w)
    assert emptied == len(f.skip) and pruned == 0
    g = q.flat()
    for k in ("mass", "is_leaf", "first", "count", "skip", "order", "path"):
        assert np.array_equal(getattr(f, k), getattr(g, k)), k
    assert np.array_equal(f.geom, g.geom, equal_nan=True)
