"""The shape check a caller's linearised tree passes before nbody_walk_tree_* lets it near the device (capi.hip,
tree_shape_ok): host only.  Valid: what the product's host builders and the oracle (bvh_tree.rs:56-158, quad_tree.rs:153-270)
produce.  Invalid: every way a foreign tree could send a stackless skip-link walk backwards, out of its arrays or into a
range that is not its children's."""
import numpy as np
import pytest


def _cloud(n, seed=5):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 2)) * 1e5).astype(np.float32), rng.integers(1, 9, n).astype(np.uint32)


def _oracle_tree(nb, orc, kind, pos, w):
    C = nb._capi
    if kind == C.TREE_BVH:
        f = orc.BVH(pos, w, leaf_size=64).flat()
        order = f.ids
    else:
        f = orc.Quad(pos, w).flat()
        order = f.order
    return dict(geom=f.geom, mass=f.mass, is_leaf=f.is_leaf, first=f.first, count=f.count, skip=f.skip, order=order, kind=kind)


@pytest.mark.parametrize("n", [0, 1, 64, 65, 5000])
@pytest.mark.parametrize("kind_name", ["BVH", "QUAD"])
def test_trees_of_the_builders_and_of_the_oracle_pass(nb, orc, kind_name, n):
    C = nb._capi
    kind = getattr(C, "TREE_" + kind_name)
    pos, w = _cloud(n)
    ok, why = C.tree_validate(C.host_tree(kind, pos, w), n)
    assert ok, why
    ok, why = C.tree_validate(_oracle_tree(nb, orc, kind, pos, w), n)
    assert ok, why


def _mutations(t, C):
    m, n = len(t["skip"]), len(t["order"])
    inner = np.flatnonzero(t["is_leaf"] == 0)
    leaves = np.flatnonzero(t["is_leaf"] != 0)
    deep = inner[len(inner) // 2]

    def mut(name, key, idx, value, expect):
        u = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
        u[key][idx] = value
        return name, u, expect

    yield mut("a skip link that points at its own node", "skip", deep, deep, "skip does not point forward")
    yield mut("a skip link that points backwards", "skip", deep, max(int(deep) - 3, 0), "skip does not point forward")
    yield mut("a skip link past the last node", "skip", m - 1, m + 1, "skip does not point forward")
    yield mut("a subtree that reaches past its parent's", "skip", deep + 1, t["skip"][deep] + 1, "past its parent")
    yield mut("a root that ends before the last node", "skip", 0, m - 1, "")
    yield mut("a leaf with nodes below it", "is_leaf", deep, 1, "leaf with nodes below")
    yield mut("a root without children", "is_leaf", leaves[0], 0, "without children")
    yield mut("a range before the particles", "first", leaves[1], -1, "range outside")
    yield mut("a range past the particles", "count", leaves[-1], n + 1, "range outside")
    yield mut("a leaf whose range overlaps its sibling's", "first", leaves[2], t["first"][leaves[2]] - 1, "")
    yield mut("an inner node whose range is not its children's", "count", deep, t["count"][deep] + 1, "")
    yield mut("a row listed twice", "order", 0, t["order"][1], "permutation")
    yield mut("a row that does not exist", "order", 0, n, "permutation")
    u = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
    u["kind"] = 7
    yield "an unknown tree kind", u, "kind"


@pytest.mark.parametrize("kind_name", ["BVH", "QUAD"])
def test_malformed_trees_are_refused_with_a_reason(nb, kind_name):
    C = nb._capi
    kind = getattr(C, "TREE_" + kind_name)
    pos, w = _cloud(5000, seed=9)
    t = C.host_tree(kind, pos, w)
    assert C.tree_validate(t)[0]
    for name, bad, expect in _mutations(t, C):
        ok, why = C.tree_validate(bad)
        assert not ok, name
        assert why and expect in why, (name, why)


def test_the_number_of_children_follows_the_tree_kind(nb):
    """A quad tree's roots have up to four children (quad_tree.rs:47-50), a BVH root exactly two (bvh_tree.rs:28): a quad
    tree offered as a BVH is refused, and so is a BVH root with a single child."""
    C = nb._capi
    pos, w = _cloud(5000, seed=11)
    q = C.host_tree(C.TREE_QUAD, pos, w)
    q["kind"] = C.TREE_BVH
    q["geom"] = np.zeros((len(q["skip"]), 6), np.float32)
    ok, why = C.tree_validate(q)
    assert not ok and "two children" in why
    # root -> one inner child -> two leaves: every range adds up, but the root has one child
    one = dict(kind=C.TREE_BVH, geom=np.zeros((4, 6), np.float32), mass=np.ones(4, np.uint32), is_leaf=np.array([0, 0, 1, 1], np.int32),
               first=np.array([0, 0, 0, 2], np.int64), count=np.array([4, 4, 2, 2], np.int64), skip=np.array([4, 4, 3, 4], np.int64),
               order=np.arange(4, dtype=np.uint32))
    ok, why = C.tree_validate(one)
    assert not ok and "two children" in why
    one["kind"] = C.TREE_QUAD
    one["geom"] = np.zeros((4, 5), np.float32)
    assert C.tree_validate(one)[0]


def test_particle_count_must_match_the_root(nb):
    C = nb._capi
    pos, w = _cloud(300, seed=13)
    t = C.host_tree(C.TREE_BVH, pos, w)
    assert C.tree_validate(t, 300)[0]
    t2 = dict(t)
    t2["order"] = np.concatenate([t["order"], np.array([300], np.uint32)])
    ok, why = C.tree_validate(t2, 301)
    assert not ok and "every particle" in why
