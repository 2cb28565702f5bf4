"""ctypes binding of include/nbody_hip.h (libnbody_hip.so).  Plain pointers and sizes only.

Loading never touches the GPU; `load()` raises if the library has not been built.  Nothing here falls back to
a CPU implementation: compute calls need a context, and `create_context` raises NBodyError when no gfx950
device is present (NBODY_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnbody_hip.so")
LAB_LIB_PATH = os.path.join(_HERE, "lib", "libnbody_hip_lab.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nbody_hip.h")

ABI_VERSION = 2   # NBODY_ABI_VERSION of include/nbody_hip.h (tests/test_capi_host.py checks the two agree)
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_DEGENERATE, ERR_NOMEM = 0, -1, -2, -3, -4, -5
ARITH_AUTO, ARITH_FAST, ARITH_EXACT = 0, 1, 2
ORDER_AS_WRITTEN, ORDER_CONSISTENT = 0, 1
TREE_BVH, TREE_QUAD = 0, 1
EXCHANGE_RCCL, EXCHANGE_PEER = 0, 1


class NBodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nbody_hip error {code}: {msg}")
        self.code = code


class Counting(C.Structure):
    """`struct Counting` (main.rs:74-79)."""
    _fields_ = [("build_bvh", C.c_double), ("sum_gravity", C.c_double), ("post_calculations", C.c_double)]


class Params(C.Structure):
    _fields_ = [("theta", C.c_float), ("clamp", C.c_float), ("leaf_size", C.c_int32), ("order", C.c_int32),
                ("arith", C.c_int32), ("quad_root_x", C.c_float), ("quad_root_y", C.c_float),
                ("quad_root_h", C.c_float)]


class TreeView(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("kind", C.c_int32), ("max_depth", C.c_int32)]


_lib = None
_vp, _i64, _i32, _f32, _f64, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_size_t

_SIGS = {
    "nbody_abi_version": (C.c_int, []),
    "nbody_create": (C.c_int, [C.POINTER(_vp), _i32]),
    "nbody_create_multi": (C.c_int, [C.POINTER(_vp), _i32, _vp]),
    "nbody_create_multi_ex": (C.c_int, [C.POINTER(_vp), _i32, _vp, _i32, _i32]),
    "nbody_multi_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    "nbody_multi_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "nbody_get_stream": (_vp, [_vp]),
    "nbody_direct_prep_dev": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _i64, _i64, _f32, _i32, _vp, _sz]),
    "nbody_direct_run_dev": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp, _f32, _f32, _i32, _i64, _i64, _vp, _sz, _vp]),
    "nbody_destroy": (None, [_vp]),
    "nbody_last_error": (C.c_char_p, [_vp]),
    "nbody_default_params": (C.c_int, [C.POINTER(Params)]),
    "nbody_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "nbody_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "nbody_upload_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "nbody_upload_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "nbody_download_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nbody_download_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nbody_num_particles": (_i64, [_vp]),
    "nbody_update_direct_f32": (C.c_int, [_vp, _f32, _i32, C.POINTER(Counting)]),
    "nbody_update_tree_f32": (C.c_int, [_vp, _i32, _f32, _i32, C.POINTER(Counting)]),
    "nbody_update_tree_f64": (C.c_int, [_vp, _i32, _f64, _i32, C.POINTER(Counting)]),
    "nbody_update_tree_async_f32": (C.c_int, [_vp, _i32, _f32, _i32]),
    "nbody_wait": (C.c_int, [_vp]),
    "nbody_update_tree_shard_f32": (C.c_int, [_vp, _i32, _f32, _i64, _i64, C.POINTER(Counting)]),
    "nbody_update_tree_shard_f64": (C.c_int, [_vp, _i32, _f64, _i64, _i64, C.POINTER(Counting)]),
    "nbody_export_slice_dev": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "nbody_import_rows_dev": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "nbody_accel_direct_f32": (C.c_int, [_vp, _vp]),
    "nbody_accel_tree_f32": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "nbody_accel_tree_f64": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "nbody_tree_info": (C.c_int, [_vp, C.POINTER(TreeView)]),
    "nbody_walk_tree_f32": (C.c_int, [_vp, _i32, _i64] + [_vp] * 7 + [_i64, _vp, _vp]),
    "nbody_walk_tree_f64": (C.c_int, [_vp, _i32, _i64] + [_vp] * 7 + [_i64, _vp, _vp]),
    "nbody_tree_validate": (C.c_int, [_i32, _i64] + [_vp] * 4 + [_i64, _vp, C.c_char_p, _sz]),
    "nbody_tree_export_f32": (C.c_int, [_vp] + [_vp] * 7),
    "nbody_tree_export_f64": (C.c_int, [_vp] + [_vp] * 7),
    "nbody_host_tree_build_f32": (C.c_int, [_i32, _i64, _vp, _vp, C.POINTER(Params), C.POINTER(_vp)]),
    "nbody_host_tree_build_f64": (C.c_int, [_i32, _i64, _vp, _vp, C.POINTER(Params), C.POINTER(_vp)]),
    "nbody_host_tree_free": (None, [_vp]),
    "nbody_host_tree_info": (C.c_int, [_vp, C.POINTER(TreeView)]),
    "nbody_host_tree_export_f32": (C.c_int, [_vp] + [_vp] * 7),
    "nbody_host_tree_export_f64": (C.c_int, [_vp] + [_vp] * 7),
    "nbody_tree_walk_stats": (C.c_int, [_vp, _i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "nbody_get_counting": (C.c_int, [_vp, C.POINTER(Counting)]),
    "nbody_direct_workspace_bytes": (_sz, [_i64, _i64]),
    "nbody_direct_step_dev": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _i64, _i64, _vp, _vp, _vp, _f32, _f32, _i32, _vp, _sz, _vp]),
    "nbody_direct_workspace_peek": (C.c_int, [_vp, _vp, C.POINTER(C.c_int32 * 4)]),
    "nbody_weights_to_mass_dev": (C.c_int, [_vp, _i64, _vp, _vp]),
    "nbody_snapshot_begin": (C.c_int, [_vp]),
    "nbody_snapshot_pending": (C.c_int, [_vp]),
    "nbody_snapshot_end_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "nbody_snapshot_end_f64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "nbody_delta_begin": (C.c_int, [_vp]),
    "nbody_delta_pending": (C.c_int, [_vp]),
    "nbody_delta_end": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint64)]),
    "nbody_delta_reset": (C.c_int, [_vp]),
    "nbody_delta_bound": (_sz, [_i64, C.c_int]),
    "nbody_delta_decoder_create": (_vp, []),
    "nbody_delta_decoder_destroy": (None, [_vp]),
    "nbody_delta_decoder_apply": (C.c_int, [_vp, _vp, _sz]),
    "nbody_delta_decoder_set_max_bodies": (C.c_int, [_vp, _i64]),
    "nbody_delta_decoder_error": (C.c_char_p, [_vp]),
    "nbody_delta_decoder_count": (_i64, [_vp]),
    "nbody_delta_decoder_is_f64": (C.c_int, [_vp]),
    "nbody_delta_decoder_step": (C.c_uint64, [_vp]),
    "nbody_delta_decoder_positions_f32": (C.c_int, [_vp, _vp]),
    "nbody_delta_decoder_positions_f64": (C.c_int, [_vp, _vp]),
    "nbody_render_rgba": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "nbody_render_rgba_dev": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp]),
    "nbody_selftest_exact_sum": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "nbody_selftest_exact_sum_f64": (C.c_int, [_vp, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nbody_selftest_exact_sum_f64_segmented": (C.c_int, [_vp, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nbody_selftest_div_pair": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "nbody_selftest_exact_sum_chunked": (C.c_int, [_vp, C.c_int64, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int64)]),
    "nbody_bvh_build_restarts": (C.c_int, [_vp]),
    "nbody_last_build_on_device": (C.c_int, [_vp]),
    "nbody_timer_create": (C.c_int, [C.POINTER(_vp)]),
    "nbody_timer_destroy": (None, [_vp]),
    "nbody_timer_read": (C.c_int, [_vp, _i32, C.POINTER(_f64), C.POINTER(_i64)]),
    "nbody_set_timer": (C.c_int, [_vp, _vp]),
}


def declared_symbols() -> list:
    """Every function include/nbody_hip.h declares (parsed from the header text)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z0-9_]+)\s*\(", txt)))


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 and load it by absolute path.  Two HIP runtimes in one
    process do not coexist (whichever comes second sees no GPU), so when PyTorch is installed its copy is loaded
    first and libnbody_hip's DT_NEEDED libamdhip64 resolves to it by SONAME.  Without PyTorch the system ROCm
    runtime is used, as for any C/Rust host."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for d in spec.submodule_search_locations:
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


_libs = {}
_active = "lab" if os.environ.get("NBODY_HIP_LIBRARY", "") == "lab" else "product"   # tools/ run with NBODY_HIP_LIBRARY=lab


def _load(which: str) -> C.CDLL:
    if which not in _libs:
        path = LAB_LIB_PATH if which == "lab" else LIB_PATH
        if not os.path.exists(path):
            raise NBodyError(ERR_NO_DEVICE, f"{path} is not built (run __graft_entry__.build() or "
                                            f"`make -C nbody-simulation_amd/csrc`); there is no CPU fallback")
        _share_hip_runtime_with_torch()
        lib = C.CDLL(path)
        lib.nbody_abi_version.restype = C.c_int
        lib.nbody_abi_version.argtypes = []
        have = lib.nbody_abi_version()
        if have != ABI_VERSION:  # before anything else is bound: a stale library would fail with an obscure AttributeError
            raise NBodyError(ERR_INVALID, f"{path} has ABI version {have}, this binding was written against {ABI_VERSION}: "
                                          f"rebuild it (`make -C nbody-simulation_amd/csrc`)")
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _libs[which] = lib
    return _libs[which]


def load() -> C.CDLL:
    """The library every call of this module goes to: libnbody_hip.so — or, inside `with laboratory():` (or with
    NBODY_HIP_LIBRARY=lab in the environment when this module is imported), libnbody_hip_lab.so."""
    global _lib
    _lib = _load(_active)
    return _lib


class laboratory:
    """`with laboratory():` — this module's calls go to libnbody_hip_lab.so (csrc/env.h: the build that honours the laboratory
    switches NBODY_DIRECT_ASM, NBODY_WALK_TILE_*, NBODY_BVH_BLIND_LEVELS, ... and contains the retired kernel variants) until the
    block ends.  Contexts made inside must be closed inside.  The product library never reads those switches."""

    def __enter__(self):
        global _active
        self.prev = _active
        _active = "lab"
        load()
        return self

    def __exit__(self, *exc):
        global _active
        _active = self.prev
        return False


def _err(ctx, code):
    msg = load().nbody_last_error(ctx)
    return NBodyError(code, msg.decode() if msg else "")


def check(ctx, code):
    if code != OK:
        raise _err(ctx, code)


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def default_params() -> Params:
    p = Params()
    check(None, load().nbody_default_params(C.byref(p)))
    return p


class Timer:
    """HIP-event timer for the dominant kernel (nbody_timer_*)."""

    def __init__(self):
        self.h = _vp()
        check(None, load().nbody_timer_create(C.byref(self.h)))

    def read(self, reset=True):
        ms, cnt = _f64(0), _i64(0)
        check(None, load().nbody_timer_read(self.h, 1 if reset else 0, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def close(self):
        if self.h:
            load().nbody_timer_destroy(self.h)
            self.h = None

    __del__ = close


def selftest_exact_sum(x, tile=4096, seq_run=64):
    """CPU emulation of the device build's exact sequential-sum scan -> (sum as np.float32, restarts)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out, st = C.c_float(0), _i64(0)
    check(None, load().nbody_selftest_exact_sum(_ptr(x) if x.size else None, x.size, int(tile), int(seq_run),
                                                C.byref(out), C.byref(st)))
    return np.float32(out.value), st.value


def selftest_exact_sum_f64(x, tile=4096, seq_run=16):
    """CPU emulation of the f64 device build's exact sequential-sum scan -> (sum as np.float64, restarts)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out, st = C.c_double(0), _i64(0)
    check(None, load().nbody_selftest_exact_sum_f64(_ptr(x) if x.size else None, x.size, int(tile), int(seq_run),
                                                    C.byref(out), C.byref(st)))
    return np.float64(out.value), st.value


def selftest_exact_sum_f64_segmented(x, seg=8192):
    """CPU emulation of the segmented f64 fold (runs per segment, predicted binades) -> (sum, runs applied)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out, used = C.c_double(0), _i64(0)
    check(None, load().nbody_selftest_exact_sum_f64_segmented(_ptr(x) if x.size else None, x.size, int(seg), C.byref(out), C.byref(used)))
    return np.float64(out.value), used.value


def selftest_div_pair(nx, ny, den, device=0):
    """csrc/div_pair.h on the device: (nx / den, ny / den) as the exact kernels compute them."""
    nx, ny, den = (np.ascontiguousarray(a, np.float32) for a in (nx, ny, den))
    assert nx.shape == ny.shape == den.shape and nx.ndim == 1
    qx, qy = np.empty_like(nx), np.empty_like(nx)
    if nx.size:
        check(None, load().nbody_selftest_div_pair(int(device), _ptr(nx), _ptr(ny), _ptr(den), nx.size, _ptr(qx), _ptr(qy)))
    return qx, qy


def selftest_exact_sum_chunked(x, chunk=2048):
    """CPU emulation of the chunked exact sum (predicted binades, runs with bounds) -> (sum, chunks taken whole)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out, used = C.c_float(0), _i64(0)
    check(None, load().nbody_selftest_exact_sum_chunked(_ptr(x) if x.size else None, x.size, int(chunk), C.byref(out), C.byref(used)))
    return np.float32(out.value), used.value


class DeltaDecoder:
    """Receiving side of the delta snapshots (host only): apply every stream since the key frame, in order."""

    def __init__(self):
        self.lib = load()
        self.h = self.lib.nbody_delta_decoder_create()
        if not self.h:
            raise MemoryError("nbody_delta_decoder_create")

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbody_delta_decoder_destroy(self.h)
            self.h = None

    __del__ = close

    def set_max_bodies(self, n: int):
        """Refuse streams whose header claims more than n bodies (the stream is untrusted input)."""
        rc = self.lib.nbody_delta_decoder_set_max_bodies(self.h, int(n))
        if rc != 0:
            raise NBodyError(rc, "delta decoder: set_max_bodies")

    def apply(self, stream: bytes):
        buf = np.frombuffer(bytes(stream), np.uint8)
        rc = self.lib.nbody_delta_decoder_apply(self.h, buf.ctypes.data_as(C.c_void_p) if buf.size else None, buf.size)
        if rc != 0:
            raise NBodyError(rc, self.lib.nbody_delta_decoder_error(self.h).decode())

    @property
    def n(self) -> int:
        return int(self.lib.nbody_delta_decoder_count(self.h))

    @property
    def step(self) -> int:
        return int(self.lib.nbody_delta_decoder_step(self.h))

    @property
    def dtype(self):
        return np.float64 if self.lib.nbody_delta_decoder_is_f64(self.h) else np.float32

    def positions(self):
        """Positions of the last applied snapshot, (n, 2), in upload (id) order."""
        n, dt = self.n, self.dtype
        if n < 0:
            raise NBodyError(ERR_INVALID, "no key frame applied yet")
        out = np.zeros((n, 2), dt)
        f = self.lib.nbody_delta_decoder_positions_f64 if dt == np.float64 else self.lib.nbody_delta_decoder_positions_f32
        rc = f(self.h, _ptr(out))
        if rc != 0:
            raise NBodyError(rc, "delta decoder: positions")
        return out


def host_tree(kind, pos, weight=None, params: "Params | None" = None):
    """Build the linearised tree on the host with the product builder (no device needed).
    -> dict(geom, mass, is_leaf, first, count, skip, order, kind, max_depth, overflow)."""
    lib = load()
    pos = np.asarray(pos)
    dt = np.float64 if pos.dtype == np.float64 else np.float32
    pos = np.ascontiguousarray(pos, dtype=dt).reshape(-1, 2)
    n = pos.shape[0]
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
    h = _vp()
    f = lib.nbody_host_tree_build_f64 if dt == np.float64 else lib.nbody_host_tree_build_f32
    rc = f(int(kind), n, _ptr(pos), _ptr(w), C.byref(params) if params is not None else None, C.byref(h))
    if rc not in (OK, ERR_DEGENERATE):
        raise _err(None, rc)
    try:
        v = TreeView()
        check(None, lib.nbody_host_tree_info(h, C.byref(v)))
        m = v.n_nodes
        out = dict(geom=np.zeros((m, 6 if v.kind == TREE_BVH else 5), dt), mass=np.zeros(m, np.uint32),
                   is_leaf=np.zeros(m, np.int32), first=np.zeros(m, np.int64), count=np.zeros(m, np.int64),
                   skip=np.zeros(m, np.int64), order=np.zeros(n, np.uint32))
        fe = lib.nbody_host_tree_export_f64 if dt == np.float64 else lib.nbody_host_tree_export_f32
        check(None, fe(h, _ptr(out["geom"]), _ptr(out["mass"]), _ptr(out["is_leaf"]), _ptr(out["first"]),
                       _ptr(out["count"]), _ptr(out["skip"]), _ptr(out["order"])))
        out.update(kind=v.kind, max_depth=v.max_depth, overflow=(rc == ERR_DEGENERATE))
        return out
    finally:
        lib.nbody_host_tree_free(h)


def _tree_arrays(tree, dt):
    return dict(geom=np.ascontiguousarray(tree["geom"], dtype=dt), mass=np.ascontiguousarray(tree["mass"], dtype=np.uint32),
                is_leaf=np.ascontiguousarray(tree["is_leaf"], dtype=np.int32), first=np.ascontiguousarray(tree["first"], dtype=np.int64),
                count=np.ascontiguousarray(tree["count"], dtype=np.int64), skip=np.ascontiguousarray(tree["skip"], dtype=np.int64),
                order=np.ascontiguousarray(tree["order"], dtype=np.uint32))


def tree_validate(tree, n_particles=None):
    """The shape check nbody_walk_tree_* applies to a caller's tree, on the host alone -> (ok, reason)."""
    lib = load()
    a = _tree_arrays(tree, np.float32)
    n = a["order"].shape[0] if n_particles is None else int(n_particles)
    buf = C.create_string_buffer(256)
    rc = lib.nbody_tree_validate(int(tree["kind"]), a["skip"].shape[0], _ptr(a["is_leaf"]), _ptr(a["first"]), _ptr(a["count"]),
                                 _ptr(a["skip"]), n, _ptr(a["order"]), buf, 256)
    return rc == OK, buf.value.decode()


class Context:
    """Owner of one nbody_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.h = _vp()
        rc = self.lib.nbody_create(C.byref(self.h), int(device))
        if rc != OK:
            self.h = None
            raise _err(None, rc)
        self.dtype = None
        self.n = 0

    @property
    def stream(self) -> int:
        """The hipStream_t the context enqueues its steps on, as an integer (torch.cuda.ExternalStream takes it)."""
        return int(self.lib.nbody_get_stream(self.h) or 0)

    def multi_info(self):
        """-> (devices, exchange or -1, chunks per direct step, bodies per target block)."""
        g, x, c, b = C.c_int(0), C.c_int(0), C.c_int(0), _i64(0)
        check(self.h, self.lib.nbody_multi_info(self.h, C.byref(g), C.byref(x), C.byref(c), C.byref(b)))
        return g.value, x.value, c.value, b.value

    def comm_count(self) -> int:
        """Ranks of the context's RCCL communicator as ncclCommCount reports them (0: none)."""
        k = C.c_int(0)
        check(self.h, self.lib.nbody_multi_comm_count(self.h, C.byref(k)))
        return k.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbody_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- params
    def get_params(self) -> Params:
        p = Params()
        check(self.h, self.lib.nbody_get_params(self.h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        check(self.h, self.lib.nbody_set_params(self.h, C.byref(p)))

    def set_timer(self, timer: "Timer | None"):
        check(self.h, self.lib.nbody_set_timer(self.h, timer.h if timer else None))

    # ---- state
    def upload(self, pos, vel, weight=None):
        pos = np.asarray(pos)
        dt = np.float64 if pos.dtype == np.float64 else np.float32
        pos = np.ascontiguousarray(pos, dtype=dt).reshape(-1, 2)
        vel = np.ascontiguousarray(vel, dtype=dt).reshape(-1, 2)
        n = pos.shape[0]
        if vel.shape[0] != n:
            raise ValueError("pos/vel length mismatch")
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.uint32)
        if w is not None and w.shape[0] != n:
            raise ValueError("weight length mismatch")
        f = self.lib.nbody_upload_f64 if dt == np.float64 else self.lib.nbody_upload_f32
        check(self.h, f(self.h, n, _ptr(pos), _ptr(vel), _ptr(w)))
        self.dtype, self.n = dt, n

    def download(self):
        """-> (pos[n,2], vel[n,2], weight[n], ids[n]) in the current row order."""
        dt, n = self.dtype, self.n
        pos = np.zeros((n, 2), dt)
        vel = np.zeros((n, 2), dt)
        w = np.zeros(n, np.uint32)
        ids = np.zeros(n, np.uint32)
        f = self.lib.nbody_download_f64 if dt == np.float64 else self.lib.nbody_download_f32
        check(self.h, f(self.h, _ptr(pos), _ptr(vel), _ptr(w), _ptr(ids)))
        return pos, vel, w, ids

    # ---- steps
    def update_direct(self, delta, n_steps=1, counter: Counting | None = None):
        check(self.h, self.lib.nbody_update_direct_f32(self.h, float(delta), int(n_steps),
                                                       C.byref(counter) if counter is not None else None))

    def update_tree(self, kind, delta, n_steps=1, counter: Counting | None = None):
        f = self.lib.nbody_update_tree_f64 if self.dtype == np.float64 else self.lib.nbody_update_tree_f32
        check(self.h, f(self.h, int(kind), float(delta), int(n_steps),
                        C.byref(counter) if counter is not None else None))

    def update_tree_async(self, kind, delta, n_steps=1):
        """f32 only: returns once the steps are enqueued; `wait()` (or any call that reads the rows) completes them."""
        check(self.h, self.lib.nbody_update_tree_async_f32(self.h, int(kind), float(delta), int(n_steps)))

    def wait(self):
        check(self.h, self.lib.nbody_wait(self.h))

    def update_tree_shard(self, kind, delta, begin, count, counter: Counting | None = None):
        f = self.lib.nbody_update_tree_shard_f64 if self.dtype == np.float64 else self.lib.nbody_update_tree_shard_f32
        check(self.h, f(self.h, int(kind), float(delta), int(begin), int(count),
                        C.byref(counter) if counter is not None else None))

    def export_slice_dev(self, begin, count, rows_ptr, pos_ptr, vel_ptr):
        check(self.h, self.lib.nbody_export_slice_dev(self.h, int(begin), int(count), _vp(rows_ptr), _vp(pos_ptr), _vp(vel_ptr)))

    def import_rows_dev(self, n_rows, rows_ptr, pos_ptr, vel_ptr):
        check(self.h, self.lib.nbody_import_rows_dev(self.h, int(n_rows), _vp(rows_ptr), _vp(pos_ptr), _vp(vel_ptr)))

    def accel_direct(self):
        acc = np.zeros((self.n, 2), np.float32)
        check(self.h, self.lib.nbody_accel_direct_f32(self.h, _ptr(acc)))
        return acc

    def accel_tree(self, kind, targets=None):
        dt = self.dtype
        f = self.lib.nbody_accel_tree_f64 if dt == np.float64 else self.lib.nbody_accel_tree_f32
        if targets is None:
            acc = np.zeros((self.n, 2), dt)
            check(self.h, f(self.h, int(kind), 0, None, _ptr(acc)))
            return acc
        tg = np.ascontiguousarray(targets, dtype=dt).reshape(-1, 2)
        acc = np.zeros_like(tg)
        check(self.h, f(self.h, int(kind), tg.shape[0], _ptr(tg), _ptr(acc)))
        return acc

    def walk_tree(self, tree, targets=None):
        """The force map alone over a caller's linearised tree (nbody_walk_tree_*): `tree` is a dict in tree_export()'s /
        host_tree()'s layout (geom, mass, is_leaf, first, count, skip, order, kind) over the uploaded particles.
        targets None: the particles themselves, in the row order after the call (BVH: tree order; quad: unchanged)."""
        dt = self.dtype
        f = self.lib.nbody_walk_tree_f64 if dt == np.float64 else self.lib.nbody_walk_tree_f32
        a = _tree_arrays(tree, dt)
        if targets is None:
            tg, nt = None, 0
            acc = np.zeros((self.n, 2), dt)
        else:
            tg = np.ascontiguousarray(targets, dtype=dt).reshape(-1, 2)
            nt = tg.shape[0]
            acc = np.zeros_like(tg)
        check(self.h, f(self.h, int(tree["kind"]), a["geom"].shape[0], _ptr(a["geom"]), _ptr(a["mass"]), _ptr(a["is_leaf"]),
                        _ptr(a["first"]), _ptr(a["count"]), _ptr(a["skip"]), _ptr(a["order"]), nt, _ptr(tg), _ptr(acc)))
        return acc

    def snapshot_begin(self):
        """Start the asynchronous hand-off of the current rows (main.rs:136-139); steps may follow at once."""
        check(self.h, self.lib.nbody_snapshot_begin(self.h))

    def snapshot_pending(self) -> bool:
        return bool(self.lib.nbody_snapshot_pending(self.h))

    def snapshot_end(self):
        """-> (position, velocity, weight, ids, steps done when the snapshot was taken)."""
        n, dt = self.n, self.dtype
        pos, vel = np.zeros((n, 2), dt), np.zeros((n, 2), dt)
        w, ids = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        step = C.c_uint64(0)
        f = self.lib.nbody_snapshot_end_f64 if dt == np.float64 else self.lib.nbody_snapshot_end_f32
        check(self.h, f(self.h, _ptr(pos), _ptr(vel), _ptr(w), _ptr(ids), C.byref(step)))
        return pos, vel, w, ids, step.value

    def delta_begin(self):
        """Encode the positions (id order) against the previous delta snapshot and start the stream's hand-off."""
        check(self.h, self.lib.nbody_delta_begin(self.h))

    def delta_pending(self) -> bool:
        return bool(self.lib.nbody_delta_pending(self.h))

    def delta_end(self, cap=None):
        """-> (the stream as bytes, steps done when it was taken)."""
        if cap is None:
            cap = int(self.lib.nbody_delta_bound(self.n, 1 if self.dtype == np.float64 else 0))
        buf = np.zeros(max(int(cap), 1), np.uint8)
        size, step = _sz(0), C.c_uint64(0)
        check(self.h, self.lib.nbody_delta_end(self.h, _ptr(buf), int(cap), C.byref(size), C.byref(step)))
        return buf[:size.value].tobytes(), step.value

    def delta_reset(self):
        check(self.h, self.lib.nbody_delta_reset(self.h))

    def render(self, height=100_000, render_px=1250):
        """The reference's draw() of the current rows -> uint8 array (render_px, render_px, 4), RGBA."""
        out = np.zeros((render_px, render_px, 4), np.uint8)
        check(self.h, self.lib.nbody_render_rgba(self.h, int(height), int(render_px), _ptr(out)))
        return out

    def last_build_on_device(self) -> bool:
        return bool(self.lib.nbody_last_build_on_device(self.h))

    def bvh_build_restarts(self) -> int:
        return int(self.lib.nbody_bvh_build_restarts(self.h))

    def tree_info(self) -> TreeView:
        v = TreeView()
        check(self.h, self.lib.nbody_tree_info(self.h, C.byref(v)))
        return v

    def tree_export(self):
        """-> dict(geom, mass, is_leaf, first, count, skip, order) of the last build."""
        v = self.tree_info()
        m, dt = v.n_nodes, self.dtype
        geom = np.zeros((m, 6 if v.kind == TREE_BVH else 5), dt)
        out = dict(geom=geom, mass=np.zeros(m, np.uint32), is_leaf=np.zeros(m, np.int32),
                   first=np.zeros(m, np.int64), count=np.zeros(m, np.int64), skip=np.zeros(m, np.int64),
                   order=np.zeros(self.n, np.uint32))
        f = self.lib.nbody_tree_export_f64 if dt == np.float64 else self.lib.nbody_tree_export_f32
        check(self.h, f(self.h, _ptr(out["geom"]), _ptr(out["mass"]), _ptr(out["is_leaf"]), _ptr(out["first"]),
                        _ptr(out["count"]), _ptr(out["skip"]), _ptr(out["order"])))
        out["kind"], out["max_depth"] = v.kind, v.max_depth
        return out

    def walk_stats(self, enable=True):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self.h, self.lib.nbody_tree_walk_stats(self.h, 1 if enable else 0, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def counting(self) -> Counting:
        c = Counting()
        check(self.h, self.lib.nbody_get_counting(self.h, C.byref(c)))
        return c


class MultiContext(Context):
    """Several GPUs of one node behind one handle (nbody_create_multi): the same calls as Context; the library shards
    every update over the devices and does the exchange itself (RCCL all-gather, or peer copies)."""

    def __init__(self, devices, exchange: int | None = None, chunks: int = 0):
        self.lib = load()
        self.h = _vp()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        if exchange is None:
            rc = self.lib.nbody_create_multi(C.byref(self.h), len(devices), ids)
        else:
            rc = self.lib.nbody_create_multi_ex(C.byref(self.h), len(devices), ids, int(exchange), int(chunks))
        if rc != OK:
            self.h = None
            raise _err(None, rc)
        self.dtype = None
        self.n = 0


def mass_hint(weight) -> float:
    """The `uniform_mass` argument of the direct step for these weights (include/nbody_hip.h): > 0 all equal,
    < 0 all equal to its magnitude but for at most n/256 bodies, 0 neither."""
    w = np.asarray(weight)
    if w.size == 0:
        return 0.0
    vals, counts = np.unique(w, return_counts=True)
    base = vals[np.argmax(counts)]
    odd = int(w.size - counts.max())
    if base <= 0:
        return 0.0
    if odd == 0:
        return float(base)
    return -float(base) if odd <= w.size // 256 else 0.0


# ---- device-pointer level (raw addresses; used with torch tensors by sharding.py / bench.py)
def direct_workspace_bytes(n_sources: int, n_targets: int) -> int:
    return int(load().nbody_direct_workspace_bytes(int(n_sources), int(n_targets)))


def direct_step_dev(stream, n_sources, pos_all, mass_all, target_begin, n_targets, vel, pos_out, acc_out, delta,
                    clamp, arith, workspace, workspace_bytes, timer: Timer | None = None, uniform_mass: float = 0.0):
    rc = load().nbody_direct_step_dev(_vp(stream), int(n_sources), _vp(pos_all), _vp(mass_all), float(uniform_mass),
                                      int(target_begin),
                                      int(n_targets), _vp(vel) if vel else None, _vp(pos_out) if pos_out else None,
                                      _vp(acc_out) if acc_out else None, float(delta), float(clamp), int(arith),
                                      _vp(workspace), int(workspace_bytes), timer.h if timer else None)
    check(None, rc)


def direct_prep_dev(stream, n_sources, pos_all, mass_all, n_targets_total, n_targets_max, clamp, arith, workspace,
                    workspace_bytes, uniform_mass: float = 0.0):
    """One preparation per step over all positions (hazard scan, near/far split); direct_run_dev per target block."""
    check(None, load().nbody_direct_prep_dev(_vp(stream), int(n_sources), _vp(pos_all), _vp(mass_all), float(uniform_mass),
                                             int(n_targets_total), int(n_targets_max), float(clamp), int(arith),
                                             _vp(workspace), int(workspace_bytes)))


def direct_run_dev(stream, n_sources, pos_all, mass_all, target_begin, n_targets, vel, pos_out, acc_out, delta, clamp,
                   arith, n_targets_total, n_targets_max, workspace, workspace_bytes, timer: Timer | None = None,
                   uniform_mass: float = 0.0):
    rc = load().nbody_direct_run_dev(_vp(stream), int(n_sources), _vp(pos_all), _vp(mass_all), float(uniform_mass),
                                     int(target_begin), int(n_targets), _vp(vel) if vel else None,
                                     _vp(pos_out) if pos_out else None, _vp(acc_out) if acc_out else None, float(delta),
                                     float(clamp), int(arith), int(n_targets_total), int(n_targets_max), _vp(workspace),
                                     int(workspace_bytes), timer.h if timer else None)
    check(None, rc)


def direct_workspace_peek(stream, workspace):
    """-> (hazard, fallback, n_near, state) of the last direct_step_dev on that workspace."""
    out = (C.c_int32 * 4)()
    check(None, load().nbody_direct_workspace_peek(_vp(stream), _vp(workspace), C.byref(out)))
    return tuple(out)


def weights_to_mass_dev(stream, n, weight_u32, mass_f32):
    check(None, load().nbody_weights_to_mass_dev(_vp(stream), int(n), _vp(weight_u32), _vp(mass_f32)))
