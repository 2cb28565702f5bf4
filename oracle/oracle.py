"""ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see nbody_oracle.hpp).

ctypes loader for the CPU restatement (oracle/nbody_oracle.cpp).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS: dict = {}

_F = {"f32": (np.float32, C.c_float), "f64": (np.float64, C.c_double)}


def build(native: bool = False, quiet: bool = True) -> str:
    """Compile the restatement with g++ (seconds).  native=True builds the -march=native flavour."""
    target = "native" if native else "all"
    subprocess.run(["make", "-C", _HERE, target], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)
    return os.path.join(_HERE, "liboracle_nbody_native.so" if native else "liboracle_nbody.so")


def lib(native: bool = False) -> C.CDLL:
    key = "native" if native else "portable"
    if key not in _LIBS:
        path = os.path.join(_HERE, "liboracle_nbody_native.so" if native else "liboracle_nbody.so")
        if not os.path.exists(path) or os.path.getmtime(path) < max(
                os.path.getmtime(os.path.join(_HERE, f)) for f in ("nbody_oracle.cpp", "nbody_oracle.hpp")):
            build(native)
        _LIBS[key] = C.CDLL(path)
        assert _LIBS[key].orc_abi_version() == 1
    return _LIBS[key]


# The clamp is the f32 constant `0.001f32` upstream (main.rs:247-248); f64 runs widen that f32 value, exactly as the
# product's `float clamp` parameter does, so every wrapper passes nt(np.float32(clamp)).  THETA is an f32 constant too
# (main.rs:35) and `float theta` in nbody_params: nt(np.float32(theta)) — for 0.5 and 50 nothing changes; for a theta like 0.7 an
# f64 run that used the double 0.7 decided one borderline node in 65 121 targets differently (found by the extended fuzz, round 3).
def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _sfx(dtype) -> str:
    return "f64" if np.dtype(dtype) == np.float64 else "f32"


def _prep(a, dtype, shape2=True):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape2:
        a = a.reshape(-1, 2)
    return a


def pair(p1, p2, force, clamp=0.001, acc=(0.0, 0.0), dtype=np.float32):
    """One evaluation of calculate_gravity (main.rs:234-253); returns the updated accumulator."""
    sfx = _sfx(dtype)
    nt, ct = _F[sfx]
    a = np.array(acc, dtype=nt)
    f = getattr(lib(), f"orc_pair_{sfx}")
    f.restype = None
    f.argtypes = [ct] * 6 + [C.POINTER(ct)]
    f(ct(nt(p1[0])), ct(nt(p1[1])), ct(nt(p2[0])), ct(nt(p2[1])), ct(nt(force)), ct(nt(np.float32(clamp))), _p(a, ct))
    return a


def direct_accel(pos, weight, targets=None, target_pos=None, clamp=0.001, accum="native", nthreads=1,
                 native_lib=False):
    """Direct O(N^2) sum.  accum='native': sequential in the particle dtype, ascending j.
    accum='f64': each term evaluated in the particle dtype as written, summed in double.
    Returns (acc[n_tgt,2] float64, norm[n_tgt] float64 = sum_j |term|_1 for accum='f64')."""
    sfx = _sfx(pos.dtype)
    nt, ct = _F[sfx]
    pos = _prep(pos, nt)
    n = pos.shape[0]
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
    idx = None
    tp = None
    if target_pos is not None:
        tp = _prep(target_pos, nt)
        nt_ = tp.shape[0]
    elif targets is not None:
        idx = np.ascontiguousarray(targets, dtype=np.int64)
        nt_ = idx.shape[0]
    else:
        nt_ = n
    acc = np.zeros((nt_, 2), dtype=np.float64)
    norm = np.zeros(nt_, dtype=np.float64)
    f = getattr(lib(native_lib), f"orc_direct_accel_{sfx}")
    f.restype = None
    f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(C.c_uint32), C.c_int64, C.POINTER(C.c_int64),
                  C.POINTER(ct), ct, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    f(n, _p(pos, ct), _p(w, C.c_uint32), nt_, _p(idx, C.c_int64), _p(tp, ct), ct(nt(np.float32(clamp))),
      0 if accum == "native" else 1, int(nthreads), _p(acc, C.c_double), _p(norm, C.c_double))
    return acc, norm


def update_direct(pos, vel, weight, delta=0.1, clamp=0.001, nsteps=1, nthreads=1, native_lib=False):
    """nsteps of direct force + semi-implicit Euler (main.rs:419-423).  Returns (pos, vel, counting[3])."""
    sfx = _sfx(pos.dtype)
    nt, ct = _F[sfx]
    pos = _prep(pos, nt).copy()
    vel = _prep(vel, nt).copy()
    w = np.ascontiguousarray(weight, dtype=np.uint32)
    cnt = np.zeros(3, dtype=np.float64)
    f = getattr(lib(native_lib), f"orc_update_direct_{sfx}")
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(ct), C.POINTER(C.c_uint32), ct, ct, C.c_int, C.c_int,
                  C.POINTER(C.c_double)]
    rc = f(pos.shape[0], _p(pos, ct), _p(vel, ct), _p(w, C.c_uint32), ct(nt(delta)), ct(nt(np.float32(clamp))),
           int(nsteps), int(nthreads), _p(cnt, C.c_double))
    if rc:
        raise RuntimeError(f"oracle update_direct rc={rc}")
    return pos, vel, cnt


AS_WRITTEN, CONSISTENT = 0, 1


def update_bvh(pos, vel, weight, delta=0.1, theta=50.0, clamp=0.001, leaf_size=64, mode=AS_WRITTEN, nsteps=1,
               nthreads=1, ids=None, native_lib=False):
    """World::update (main.rs:388-425).  Arrays come back in the (permuted) order the reference would hold
    them in; `ids` carries each row's original index.  Returns (pos, vel, weight, ids, counting[3])."""
    sfx = _sfx(pos.dtype)
    nt, ct = _F[sfx]
    pos = _prep(pos, nt).copy()
    vel = _prep(vel, nt).copy()
    w = np.ascontiguousarray(weight, dtype=np.uint32).copy()
    ids = np.arange(pos.shape[0], dtype=np.uint32) if ids is None else np.ascontiguousarray(ids, np.uint32).copy()
    cnt = np.zeros(3, dtype=np.float64)
    f = getattr(lib(native_lib), f"orc_update_bvh_{sfx}")
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(ct), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), ct, ct,
                  ct, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    rc = f(pos.shape[0], _p(pos, ct), _p(vel, ct), _p(w, C.c_uint32), _p(ids, C.c_uint32), ct(nt(delta)),
           ct(nt(np.float32(theta))), ct(nt(np.float32(clamp))), int(leaf_size), int(mode), int(nsteps), int(nthreads),
           _p(cnt, C.c_double))
    if rc:
        raise RuntimeError(f"oracle update_bvh rc={rc} (degenerate input: recursion depth cap)")
    return pos, vel, w, ids, cnt


def update_quad(pos, vel, weight, delta=0.1, theta=50.0, clamp=0.001, root=(0.0, 0.0, 100000.0), nsteps=1,
                nthreads=1, native_lib=False):
    sfx = _sfx(pos.dtype)
    nt, ct = _F[sfx]
    pos = _prep(pos, nt).copy()
    vel = _prep(vel, nt).copy()
    w = np.ascontiguousarray(weight, dtype=np.uint32)
    cnt = np.zeros(3, dtype=np.float64)
    f = getattr(lib(native_lib), f"orc_update_quad_{sfx}")
    f.restype = C.c_int
    f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(ct), C.POINTER(C.c_uint32), ct, ct, ct, ct, ct, ct,
                  C.c_int, C.c_int, C.POINTER(C.c_double)]
    rc = f(pos.shape[0], _p(pos, ct), _p(vel, ct), _p(w, C.c_uint32), ct(nt(delta)), ct(nt(np.float32(theta))),
           ct(nt(np.float32(clamp))), ct(nt(root[0])), ct(nt(root[1])), ct(nt(root[2])), int(nsteps), int(nthreads),
           _p(cnt, C.c_double))
    if rc:
        raise RuntimeError(f"oracle update_quad rc={rc} (degenerate input: recursion depth cap)")
    return pos, vel, cnt


def draw(pos, vel, weight, height=100_000, render_px=1250):
    """draw() of main.rs:41-72 over the rows in the order given -> uint8 (render_px, render_px, 4)."""
    pos = np.asarray(pos)
    dtype = np.float64 if pos.dtype == np.float64 else np.float32
    ct = C.c_double if dtype == np.float64 else C.c_float
    pos = _prep(pos, dtype)
    vel = _prep(vel, dtype)
    w = np.ascontiguousarray(weight, dtype=np.uint32)
    out = np.zeros((render_px, render_px, 4), np.uint8)
    f = getattr(lib(), "orc_draw_" + _sfx(dtype))
    f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(ct), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
    f.restype = None
    f(pos.shape[0], _p(pos, ct), _p(vel, ct), _p(w, C.c_uint32), int(height), int(render_px), _p(out, C.c_uint8))
    return out


@dataclass
class FlatBVH:
    geom: np.ndarray      # [n_nodes, 6] off_x off_y size_x size_y cog_x cog_y
    mass: np.ndarray      # u32
    is_leaf: np.ndarray   # i32
    first: np.ndarray     # i64 (leaf slice into the permuted particle array)
    count: np.ndarray
    skip: np.ndarray      # i64 pre-order index after this subtree
    pos_perm: np.ndarray  # [n,2] particle positions after the in-place partitioning
    ids: np.ndarray       # u32 original index of each permuted row
    overflow: bool


class BVH:
    """Handle on a built reference BVH (bvh_tree.rs:56-158) over `pos`."""

    def __init__(self, pos, weight=None, leaf_size=64):
        self.sfx = _sfx(pos.dtype)
        self.nt, self.ct = _F[self.sfx]
        pos = _prep(pos, self.nt)
        self.n = pos.shape[0]
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
        L = lib()
        f = getattr(L, f"orc_bvh_create_{self.sfx}")
        f.restype = C.c_void_p
        f.argtypes = [C.c_int64, C.POINTER(self.ct), C.POINTER(C.c_uint32), C.c_int64]
        self.h = C.c_void_p(f(self.n, _p(pos, self.ct), _p(w, C.c_uint32), int(leaf_size)))
        self._L = L

    def close(self):
        if self.h:
            f = getattr(self._L, f"orc_bvh_free_{self.sfx}")
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(self.h)
            self.h = None

    __del__ = close

    def flat(self) -> FlatBVH:
        L, s, ct = self._L, self.sfx, self.ct
        fn = getattr(L, f"orc_bvh_num_nodes_{s}")
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p]
        m = fn(self.h)
        fo = getattr(L, f"orc_bvh_overflow_{s}")
        fo.restype = C.c_int
        fo.argtypes = [C.c_void_p]
        geom = np.zeros((m, 6), self.nt)
        mass = np.zeros(m, np.uint32)
        is_leaf = np.zeros(m, np.int32)
        first = np.zeros(m, np.int64)
        count = np.zeros(m, np.int64)
        skip = np.zeros(m, np.int64)
        pos_perm = np.zeros((self.n, 2), self.nt)
        ids = np.zeros(self.n, np.uint32)
        fe = getattr(L, f"orc_bvh_export_{s}")
        fe.restype = None
        fe.argtypes = [C.c_void_p, C.POINTER(ct), C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(ct),
                       C.POINTER(C.c_uint32)]
        fe(self.h, _p(geom, ct), _p(mass, C.c_uint32), _p(is_leaf, C.c_int32), _p(first, C.c_int64),
           _p(count, C.c_int64), _p(skip, C.c_int64), _p(pos_perm, ct), _p(ids, C.c_uint32))
        return FlatBVH(geom, mass, is_leaf, first, count, skip, pos_perm, ids, bool(fo(self.h)))

    def walk(self, targets, theta=50.0, clamp=0.001, nthreads=1, stats=False):
        """bvh_sum_gravity (main.rs:348-386) for each target position.  stats=True also returns
        (node_visits, accepted, leaf_pairs) summed over the targets."""
        ct = self.ct
        tg = _prep(targets, self.nt)
        acc = np.zeros_like(tg)
        st = np.zeros(3, np.uint64) if stats else None
        f = getattr(self._L, f"orc_bvh_walk_{self.sfx}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ct), ct, ct, C.c_int, C.POINTER(ct), C.POINTER(C.c_uint64)]
        f(self.h, tg.shape[0], _p(tg, ct), ct(self.nt(np.float32(theta))), ct(self.nt(np.float32(clamp))), int(nthreads), _p(acc, ct),
          _p(st, C.c_uint64))
        return (acc, st) if stats else acc

    def walk_ref(self, targets, theta=50.0, clamp=0.001, nthreads=1):
        """The tolerance reference of the walk: the same interaction list, every term as main.rs:252 writes it in the tree's
        precision, accumulated in double -> (acc64[n,2], norm[n] = sum |term|_1)."""
        ct = self.ct
        tg = _prep(targets, self.nt)
        acc = np.zeros(tg.shape, np.float64)
        norm = np.zeros(tg.shape[0], np.float64)
        f = getattr(self._L, f"orc_bvh_walk_ref_{self.sfx}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ct), ct, ct, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        f(self.h, tg.shape[0], _p(tg, ct), ct(self.nt(np.float32(theta))), ct(self.nt(np.float32(clamp))), int(nthreads),
          _p(acc, C.c_double), _p(norm, C.c_double))
        return acc, norm


@dataclass
class FlatQuad:
    geom: np.ndarray        # [n_nodes, 5] off_x off_y height cog_x cog_y
    mass: np.ndarray
    is_leaf: np.ndarray
    depth: np.ndarray
    child_code: np.ndarray
    path: np.ndarray        # u64: 2-bit child codes from the root
    first: np.ndarray
    count: np.ndarray
    skip: np.ndarray
    order: np.ndarray       # u32 particle ids in leaf (DFS) order
    overflow: bool


class Quad:
    """Handle on a quad tree built by quad_tree.rs:153-270 semantics, particles inserted in index order."""

    def __init__(self, pos, weight=None, root=(0.0, 0.0, 100000.0)):
        self.sfx = _sfx(pos.dtype)
        self.nt, self.ct = _F[self.sfx]
        pos = _prep(pos, self.nt)
        self.n = pos.shape[0]
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
        L = lib()
        ct = self.ct
        f = getattr(L, f"orc_quad_create_{self.sfx}")
        f.restype = C.c_void_p
        f.argtypes = [C.c_int64, C.POINTER(ct), C.POINTER(C.c_uint32), ct, ct, ct]
        self.h = C.c_void_p(f(self.n, _p(pos, ct), _p(w, C.c_uint32), ct(self.nt(root[0])), ct(self.nt(root[1])),
                              ct(self.nt(root[2]))))
        self._L = L

    def close(self):
        if self.h:
            f = getattr(self._L, f"orc_quad_free_{self.sfx}")
            f.restype = None
            f.argtypes = [C.c_void_p]
            f(self.h)
            self.h = None

    __del__ = close

    def flat(self) -> FlatQuad:
        L, s, ct = self._L, self.sfx, self.ct
        fn = getattr(L, f"orc_quad_num_nodes_{s}")
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p]
        m = fn(self.h)
        fo = getattr(L, f"orc_quad_overflow_{s}")
        fo.restype = C.c_int
        fo.argtypes = [C.c_void_p]
        geom = np.zeros((m, 5), self.nt)
        mass = np.zeros(m, np.uint32)
        is_leaf = np.zeros(m, np.int32)
        depth = np.zeros(m, np.int32)
        code = np.zeros(m, np.uint32)
        path = np.zeros(m, np.uint64)
        first = np.zeros(m, np.int64)
        count = np.zeros(m, np.int64)
        skip = np.zeros(m, np.int64)
        order = np.zeros(self.n, np.uint32)
        fe = getattr(L, f"orc_quad_export_{s}")
        fe.restype = None
        fe.argtypes = [C.c_void_p, C.POINTER(ct), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                       C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]
        fe(self.h, _p(geom, ct), _p(mass, C.c_uint32), _p(is_leaf, C.c_int32), _p(depth, C.c_int32),
           _p(code, C.c_uint32), _p(path, C.c_uint64), _p(first, C.c_int64), _p(count, C.c_int64),
           _p(skip, C.c_int64), _p(order, C.c_uint32))
        return FlatQuad(geom, mass, is_leaf, depth, code, path, first, count, skip, order, bool(fo(self.h)))

    def reuse(self, pos, weight=None):
        """The kept tree's next step (quad_tree.rs:66-137; no caller upstream): empty(), the points inserted where they are
        now, calculate_gravity(), prune() -> (empty()'s return, prune()'s return).  flat() / walk() then see the kept tree."""
        pos = _prep(pos, self.nt)
        self.n = pos.shape[0]
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
        counts = np.zeros(2, np.uint32)
        f = getattr(self._L, f"orc_quad_reuse_{self.sfx}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(self.ct), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        f(self.h, self.n, _p(pos, self.ct), _p(w, C.c_uint32), _p(counts, C.c_uint32))
        return int(counts[0]), int(counts[1])

    def empty(self):
        """QuadTree::empty alone (quad_tree.rs:66-89) -> the cells visited.  (flat() is stale until reuse().)"""
        f = getattr(self._L, f"orc_quad_empty_{self.sfx}")
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def walk(self, targets, theta=50.0, clamp=0.001, nthreads=1, stats=False):
        ct = self.ct
        tg = _prep(targets, self.nt)
        acc = np.zeros_like(tg)
        st = np.zeros(3, np.uint64) if stats else None
        f = getattr(self._L, f"orc_quad_walk_{self.sfx}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ct), ct, ct, C.c_int, C.POINTER(ct), C.POINTER(C.c_uint64)]
        f(self.h, tg.shape[0], _p(tg, ct), ct(self.nt(np.float32(theta))), ct(self.nt(np.float32(clamp))), int(nthreads), _p(acc, ct),
          _p(st, C.c_uint64))
        return (acc, st) if stats else acc

    def walk_ref(self, targets, theta=50.0, clamp=0.001, nthreads=1):
        """The tolerance reference of the walk: the same interaction list, every term as main.rs:252 writes it in the tree's
        precision, accumulated in double -> (acc64[n,2], norm[n] = sum |term|_1)."""
        ct = self.ct
        tg = _prep(targets, self.nt)
        acc = np.zeros(tg.shape, np.float64)
        norm = np.zeros(tg.shape[0], np.float64)
        f = getattr(self._L, f"orc_quad_walk_ref_{self.sfx}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ct), ct, ct, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        f(self.h, tg.shape[0], _p(tg, ct), ct(self.nt(np.float32(theta))), ct(self.nt(np.float32(clamp))), int(nthreads),
          _p(acc, C.c_double), _p(norm, C.c_double))
        return acc, norm
