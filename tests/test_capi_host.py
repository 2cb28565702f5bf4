"""CPU-side checks of the product: the C-ABI library loads, exports every declared symbol, refuses to run without
a device, and its host-side tree builders reproduce the oracle's trees bit for bit.  No GPU, no compute calls."""
import ctypes
import os

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(nb):
    C = nb._capi
    lib = ctypes.CDLL(C.LIB_PATH)
    declared = C.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # and the ctypes table binds exactly the declared set
    assert sorted(C._SIGS) == declared


def _env_names_in(path):
    """Every standalone string NBODY_[A-Z0-9_]+ in a binary: what it can pass to getenv (names inside messages do not count)."""
    import re
    with open(path, "rb") as f:
        blob = f.read()
    return {m.group(1).decode() for m in re.finditer(rb"\x00(NBODY_[A-Z0-9_]+)\x00", blob)}


def test_product_library_reads_only_the_documented_environment(nb):
    """a host that links libnbody_hip.so gets one behaviour — the library reads the nine variables that
    include/nbody_hip.h documents and no others; the A/B switches and test hooks live in the laboratory build only
    (csrc/env.h: lab_int / lab_str compile to their defaults in the product)."""
    import re
    C = nb._capi
    with open(C.HEADER_PATH) as f:
        head = f.read().split("#ifndef NBODY_HIP_H")[0]
    env_block = head[head.index("Environment."):]
    documented = set(re.findall(r"^ \*   (NBODY_[A-Z0-9_]+)", env_block, flags=re.M))
    assert 5 <= len(documented) <= 10, documented
    in_product = _env_names_in(C.LIB_PATH)
    assert in_product == documented, (sorted(in_product - documented), sorted(documented - in_product))
    # the laboratory build is where the rest went: same symbols, the switches on top
    assert os.path.exists(C.LAB_LIB_PATH), "make -C nbody-simulation_amd/csrc builds both"
    in_lab = _env_names_in(C.LAB_LIB_PATH)
    assert len(in_lab) > 25 and {"NBODY_DIRECT_ASM", "NBODY_WALK_TILE_POISON", "NBODY_BVH_BLIND_LEVELS"} <= in_lab
    lab = ctypes.CDLL(C.LAB_LIB_PATH)
    assert not [s for s in C.declared_symbols() if not hasattr(lab, s)]
    # and the retired kernel variants are not in the product's device code: the per-thread walk, the three-pass walk
    with open(C.LIB_PATH, "rb") as f:
        blob = f.read()
    for retired in (b"walk_pass", b"walk_sum", b"9tree_walkI"):   # (mangled nbody::tree_walk<...>; tree_walk_wave / _small stay)
        assert retired not in blob, retired
    with open(C.LAB_LIB_PATH, "rb") as f:
        assert b"walk_pass" in f.read()


def test_abi_version_and_defaults(nb):
    C = nb._capi
    import re
    with open(C.HEADER_PATH) as f:
        declared = int(re.search(r"#define\s+NBODY_ABI_VERSION\s+(\d+)", f.read()).group(1))
    assert C.load().nbody_abi_version() == declared == C.ABI_VERSION
    p = C.default_params()
    # the reference's constants: THETA main.rs:35, clamp :247, TARGET_POINTS bvh_tree.rs:37, HEIGHT main.rs:31
    assert (p.theta, p.leaf_size, p.quad_root_h) == (50.0, 64, 100000.0)
    assert p.clamp == np.float32(0.001)
    assert p.order == C.ORDER_AS_WRITTEN and p.arith == C.ARITH_AUTO


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_cpu_fallback(nb):
    with pytest.raises(nb._capi.NBodyError) as e:
        nb._capi.Context(0)
    assert e.value.code == nb._capi.ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_multi_context_has_no_cpu_fallback_either(nb):
    """nbody_create_multi: no device -> NBODY_ERR_NO_DEVICE (the first single-device context it makes says so); bad
    arguments are refused before any device is touched."""
    C = nb._capi
    for exchange in (None, C.EXCHANGE_RCCL, C.EXCHANGE_PEER):
        with pytest.raises(C.NBodyError) as e:
            C.MultiContext([0], exchange)
        assert e.value.code == C.ERR_NO_DEVICE and "no CPU path" in str(e.value)
    for bad in ([], list(range(65))):
        with pytest.raises(C.NBodyError) as e:
            C.MultiContext(bad, C.EXCHANGE_PEER)
        assert e.value.code == C.ERR_INVALID
    with pytest.raises(C.NBodyError) as e:
        C.MultiContext([0, 0], C.EXCHANGE_RCCL)          # one rank per physical device under RCCL
    assert e.value.code == C.ERR_INVALID and "one rank per physical device" in str(e.value)
    with pytest.raises(C.NBodyError) as e:
        C.MultiContext([0], 7)
    assert e.value.code == C.ERR_INVALID
    with pytest.raises(C.NBodyError) as e:
        C.MultiContext([0], C.EXCHANGE_PEER, chunks=99)
    assert e.value.code == C.ERR_INVALID


def test_mass_hint_of_the_direct_step(nb):
    from nbody_simulation_amd._capi import mass_hint
    assert mass_hint(np.ones(1000, np.uint32)) == 1.0
    w = np.ones(151409, np.uint32)
    w[0], w[1] = 75_000_000, 750_000
    assert mass_hint(w) == -1.0                                  # World::new: two heavy bodies (main.rs:282-291)
    assert mass_hint((np.arange(1000) % 3 + 1).astype(np.uint32)) == 0.0
    assert mass_hint(np.zeros(8, np.uint32)) == 0.0              # a zero base mass is no hint
    assert mass_hint(np.zeros(0, np.uint32)) == 0.0


def test_workspace_size_is_monotone_and_small(nb):
    C = nb._capi
    a = C.direct_workspace_bytes(1 << 20, 1 << 20)
    b = C.direct_workspace_bytes(1 << 20, 1 << 17)
    assert 256 <= a <= 128 << 20 and 256 <= b <= 128 << 20


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,leaf", [(0, 64), (1, 64), (40, 64), (64, 64), (65, 64), (1024, 64), (777, 16), (50000, 64)])
def test_host_bvh_matches_oracle_bit_exact(nb, orc, dtype, n, leaf):
    """Tree node indexing bit-exact: the product's iterative pre-order builder vs the oracle's recursive
    restatement of bvh_tree.rs:56-158."""
    C = nb._capi
    pos, _, _ = nb.scenes.plummer(n, seed=31)
    pos = pos.astype(dtype)
    w = (np.arange(n) % 9 + 1).astype(np.uint32)
    prm = C.default_params()
    prm.leaf_size = leaf
    t = C.host_tree(C.TREE_BVH, pos, w, prm)
    o = orc.BVH(pos, w, leaf_size=leaf).flat()
    assert not t["overflow"] and not o.overflow
    for k in ("mass", "is_leaf", "first", "count", "skip"):
        assert np.array_equal(t[k], getattr(o, k)), k
    assert np.array_equal(t["geom"], o.geom, equal_nan=True)
    assert np.array_equal(t["order"], o.ids)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [0, 1, 8, 9, 100, 1024, 50000])
def test_host_quad_matches_oracle_bit_exact(nb, orc, dtype, n):
    """The product builds the quad tree top-down with a stable 4-way split; the oracle inserts point by point as
    quad_tree.rs:153-227 does.  Cells, child codes, leaf orders, masses and centres of gravity must be identical."""
    C = nb._capi
    pos, _, _ = nb.scenes.plummer(n, seed=32)
    pos = pos.astype(dtype)
    w = (np.arange(n) % 4 + 1).astype(np.uint32)
    t = C.host_tree(C.TREE_QUAD, pos, w)
    o = orc.Quad(pos, w).flat()
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(t[k], getattr(o, k)), k


def test_host_trees_mixed_masses_and_heavy_bodies(nb, orc):
    """The reference's own scene has masses 1, 750 000 and 75 000 000 (main.rs:282-291)."""
    C = nb._capi
    pos, _, w = nb.scenes.galaxy()
    pos, w = pos[:30000], w[:30000]
    t = C.host_tree(C.TREE_BVH, pos, w)
    o = orc.BVH(pos, w).flat()
    assert np.array_equal(t["geom"], o.geom) and np.array_equal(t["mass"], o.mass) and np.array_equal(t["order"], o.ids)
    assert t["mass"][0] == (int(w.astype(np.uint64).sum()) & 0xFFFFFFFF)


def test_host_tree_degenerate_inputs_report_not_hang(nb):
    C = nb._capi
    pos = np.tile(np.array([[5.0, 5.0]], np.float32), (100, 1))
    assert C.host_tree(C.TREE_BVH, pos)["overflow"]
    assert C.host_tree(C.TREE_QUAD, pos)["overflow"]
    # > 8 points beyond the same corner of the root cell can never be separated either (no bounds check upstream)
    far = np.tile(np.array([[2e5, 2e5]], np.float32), (9, 1)) + np.arange(9, dtype=np.float32)[:, None]
    assert C.host_tree(C.TREE_QUAD, far)["overflow"]
    # 64 coincident points still fit one BVH leaf; 8 fit one quad leaf
    assert not C.host_tree(C.TREE_BVH, pos[:64])["overflow"]
    assert not C.host_tree(C.TREE_QUAD, pos[:8])["overflow"]


def test_host_quad_points_outside_root_cell(nb, orc):
    """insert() has no bounds check (quad_tree.rs:153-207): outside points go to edge cells by comparison only."""
    C = nb._capi
    rng = np.random.default_rng(5)
    inside = (rng.random((500, 2)) * 1e5).astype(np.float32)
    outside = np.array([[-5e4, 2e4], [1.7e5, 3e4], [4e4, -1e3], [5e4, 2.5e5], [-1.0, -1.0], [1e5 + 1, 1e5 + 1]], np.float32)
    pos = np.concatenate([inside[:250], outside, inside[250:]])
    t = C.host_tree(C.TREE_QUAD, pos)
    o = orc.Quad(pos).flat()
    assert not t["overflow"] and not o.overflow
    for k in ("geom", "mass", "is_leaf", "first", "count", "skip", "order"):
        assert np.array_equal(t[k], getattr(o, k)), k


# ------------------------------------------------------------------ exact sequential-sum scan (csrc/exact_sum.h)
def _seq_sum_f32(x):
    """The chain `sum = sum + x[i]` of bvh_tree.rs:58-61 in f32 (np.cumsum accumulates left to right in the dtype)."""
    x = np.asarray(x, np.float32)
    if x.size == 0:
        return np.float32(0)
    with np.errstate(all="ignore"):
        return np.cumsum(x, dtype=np.float32)[-1]


def _bits(v):
    return np.asarray(v, np.float32).view(np.uint32)


@pytest.mark.parametrize("tile,seq_run", [(4096, 64), (1024, 64), (256, 64), (4, 1), (7, 3)])
def test_exact_sum_scan_reproduces_the_sequential_chain(nb, tile, seq_run):
    """The parallel formulation the device BVH build uses for `sum / len` must round exactly like the plain loop:
    same-signed data, mixed signs (chains through zero), ties, subnormals, huge dynamic range, inf/NaN, overflow."""
    C = nb._capi
    rng = np.random.default_rng(99)
    cases = {}
    for n in (0, 1, 2, 63, 64, 65, 1000, 151405):
        cases[f"uniform{n}"] = rng.random(n) * 1e5
        cases[f"negative{n}"] = -rng.random(n) * 1e5
        cases[f"centred{n}"] = rng.standard_normal(n) * 3e4
        cases[f"halves{n}"] = rng.integers(0, 2000, n) * 0.5
        cases[f"halves_pm{n}"] = rng.integers(-1000, 1000, n) * 0.5
        cases[f"wide{n}"] = 10.0 ** rng.uniform(-12, 12, n) * rng.choice([-1.0, 1.0], n)
        cases[f"ones{n}"] = np.ones(n)
        cases[f"pow2{n}"] = 2.0 ** rng.integers(-30, 30, n)
    x = rng.random(5000) * 1e5
    x[::97] = 1e-40
    cases["subnormals"] = x
    for name, bad in (("inf", np.inf), ("nan", np.nan), ("big", 3e38)):
        x = rng.random(5000) * 1e5
        x[2500] = bad
        x[2501] = bad
        cases[name] = x
    for name, x in cases.items():
        x = np.asarray(x, np.float32)
        got, _ = C.selftest_exact_sum(x, tile, seq_run)
        want = _seq_sum_f32(x)
        assert _bits(got) == _bits(want) or (np.isnan(got) and np.isnan(want)), (name, got, want)


def test_exact_sum_scan_random_bit_patterns(nb):
    C = nb._capi
    rng = np.random.default_rng(3)
    for it in range(3000):
        n = int(rng.integers(1, 48))
        b = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        mode = it % 3
        if mode == 1:   # exponents clustered so that chains stay inside a few binades
            b = (b & np.uint32(0x807FFFFF)) | (rng.integers(110, 150, n).astype(np.uint32) << np.uint32(23))
        elif mode == 2:  # few mantissa bits: many exact ties
            b &= np.uint32(0xFFF80000)
        x = b.view(np.float32)
        got, _ = C.selftest_exact_sum(x, 4 if it % 2 else 64, 1 + it % 3)
        want = _seq_sum_f32(x)
        assert _bits(got) == _bits(want) or (np.isnan(got) and np.isnan(want)), (it, x, got, want)


def _seq_sum_f64(x):
    x = np.asarray(x, np.float64)
    if x.size == 0:
        return np.float64(0)
    with np.errstate(all="ignore"):
        return np.cumsum(x, dtype=np.float64)[-1]


def _bits64(v):
    return np.asarray(v, np.float64).view(np.uint64)


@pytest.mark.parametrize("tile,seq_run", [(4096, 16), (1024, 16), (4, 1), (7, 3)])
def test_exact_sum_scan_f64_reproduces_the_sequential_chain(nb, tile, seq_run):
    """The f64 twin (csrc/exact_sum64.h) that the device build of f64 BVHs runs: 53-bit significands, 64-bit increments."""
    C = nb._capi
    rng = np.random.default_rng(199)
    cases = {}
    for n in (0, 1, 2, 63, 64, 65, 1000, 200003):
        cases[f"uniform{n}"] = rng.random(n) * 1e5
        cases[f"negative{n}"] = -rng.random(n) * 1e5
        cases[f"centred{n}"] = rng.standard_normal(n) * 3e4
        cases[f"f32grid{n}"] = (rng.random(n) * 1e5).astype(np.float32).astype(np.float64)    # few mantissa bits: ties
        cases[f"halves_pm{n}"] = rng.integers(-1000, 1000, n) * 0.5
        cases[f"wide{n}"] = 10.0 ** rng.uniform(-100, 100, n) * rng.choice([-1.0, 1.0], n)
        cases[f"pow2{n}"] = 2.0 ** rng.integers(-60, 60, n)
        cases[f"tiny_ulps{n}"] = 1.0 + rng.integers(-3, 4, n) * 2.0 ** -53                  # ties at every add
    x = rng.random(5000) * 1e5
    x[::97] = 1e-310
    cases["subnormals"] = x
    for name, bad in (("inf", np.inf), ("nan", np.nan), ("big", 1.7e308)):
        x = rng.random(5000) * 1e5
        x[2500] = bad
        x[2501] = bad
        cases[name] = x
    for name, x in cases.items():
        x = np.asarray(x, np.float64)
        got, _ = C.selftest_exact_sum_f64(x, tile, seq_run)
        want = _seq_sum_f64(x)
        assert _bits64(got) == _bits64(want) or (np.isnan(got) and np.isnan(want)), (name, got, want)


@pytest.mark.parametrize("seg", [8192, 64, 8])
def test_segmented_exact_sum_f64_reproduces_the_sequential_chain(nb, seg):
    """Long f64 chains are cut into segments whose runs are prepared for a predicted binade; a run may only be used when
    the prediction and its bounds hold for the true state, so the result is the plain loop's whatever the data — and on
    ordinary coordinates nearly every segment's run is used."""
    C = nb._capi
    rng = np.random.default_rng(141)
    cases = {}
    for n in (0, 1, 65, 5000, 200003):
        cases[f"uniform{n}"] = rng.random(n) * 1e5
        cases[f"negative{n}"] = -rng.random(n) * 1e5
        cases[f"centred{n}"] = rng.standard_normal(n) * 3e4
        cases[f"f32grid{n}"] = (rng.random(n) * 1e5).astype(np.float32).astype(np.float64)
        cases[f"halves_pm{n}"] = rng.integers(-1000, 1000, n) * 0.5
        cases[f"wide{n}"] = 10.0 ** rng.uniform(-100, 100, n) * rng.choice([-1.0, 1.0], n)
        cases[f"tiny_ulps{n}"] = 1.0 + rng.integers(-3, 4, n) * 2.0 ** -53
    for name, bad in (("inf", np.inf), ("nan", np.nan), ("big", 1.7e308)):
        x = rng.random(50000) * 1e5
        x[25000] = bad
        cases[name] = x
    for name, x in cases.items():
        x = np.asarray(x, np.float64)
        got, used = C.selftest_exact_sum_f64_segmented(x, seg)
        want = _seq_sum_f64(x)
        assert _bits64(got) == _bits64(want) or (np.isnan(got) and np.isnan(want)), (name, got, want)
        if name == "uniform200003":
            assert used >= (200003 // seg) * 0.7, used        # all but the segments in which the sum crosses a power of two


def test_exact_sum_scan_f64_random_bit_patterns(nb):
    C = nb._capi
    rng = np.random.default_rng(5)
    for it in range(3000):
        n = int(rng.integers(1, 48))
        b = rng.integers(0, 2**64, n, dtype=np.uint64)
        mode = it % 3
        if mode == 1:   # exponents clustered so that chains stay inside a few binades
            b = (b & np.uint64(0x800FFFFFFFFFFFFF)) | (rng.integers(1000, 1050, n).astype(np.uint64) << np.uint64(52))
        elif mode == 2:  # few mantissa bits: many exact ties
            b &= np.uint64(0xFFFFFF0000000000)
        x = b.view(np.float64)
        got, _ = C.selftest_exact_sum_f64(x, 4 if it % 2 else 64, 1 + it % 3)
        want = _seq_sum_f64(x)
        assert _bits64(got) == _bits64(want) or (np.isnan(got) and np.isnan(want)), (it, x, got, want)


def test_exact_sum_scan_restarts_are_rare_on_scene_data(nb):
    """Same-signed coordinates leave a binade about once per doubling of the sum: the scan restarts O(log) times."""
    C = nb._capi
    pos, _, _ = nb.scenes.galaxy()
    for col in (0, 1):
        got, restarts = C.selftest_exact_sum(pos[:, col], 4096, 64)
        assert _bits(got) == _bits(_seq_sum_f32(pos[:, col]))
        assert restarts <= 40


@pytest.mark.parametrize("chunk", [2048, 64, 8])
def test_chunked_exact_sum_reproduces_the_sequential_chain(nb, chunk):
    """Long chains are cut into chunks whose runs are prepared for a predicted binade; a run may only be used when the
    prediction and its bounds hold, so the result is the plain loop's whatever the data."""
    C = nb._capi
    rng = np.random.default_rng(41)
    cases = {}
    for n in (0, 1, 65, 5000, 151405):
        cases[f"uniform{n}"] = rng.random(n) * 1e5
        cases[f"negative{n}"] = -rng.random(n) * 1e5
        cases[f"centred{n}"] = rng.standard_normal(n) * 3e4
        cases[f"lattice{n}"] = rng.integers(0, 7000, n) * 14.0
        cases[f"halves_pm{n}"] = rng.integers(-1000, 1000, n) * 0.5
        cases[f"wide{n}"] = 10.0 ** rng.uniform(-12, 12, n) * rng.choice([-1.0, 1.0], n)
    for name, bad in (("inf", np.inf), ("nan", np.nan), ("big", 3e38)):
        x = rng.random(20000) * 1e5
        x[9000] = bad
        x[9001] = bad
        cases[name] = x
    for name, x in cases.items():
        x = np.asarray(x, np.float32)
        got, used = C.selftest_exact_sum_chunked(x, chunk)
        want = _seq_sum_f32(x)
        assert _bits(got) == _bits(want) or (np.isnan(got) and np.isnan(want)), (name, got, want)
        if name == "uniform151405" and chunk == 2048:
            assert used >= 70      # every chunk but the first, and both halves of the chunks a power of two falls into


def test_header_is_c99_and_a_c_program_links_the_library(tmp_path):
    """The boundary is a C ABI: the header compiles as pedantic C99 and as C++17, and a plain C caller — what a Rust `extern "C"`
    block amounts to — links the product library and runs its host-only entry points (no device is touched)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc, libdir = os.path.join(root, "include"), os.path.join(root, "nbody-simulation_amd", "lib")
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "nbody_hip.h"
int main(void) {
  nbody_params p;
  if (nbody_abi_version() < 2 || nbody_default_params(&p) != NBODY_OK) return 1;
  if (p.theta != 50.0f || p.clamp != 0.001f || p.leaf_size != 64) return 2;       /* main.rs:35, :247-248, bvh_tree.rs:37 */
  /* a root with two leaves over three particles ... */
  int32_t is_leaf[3] = {0, 1, 1};
  int64_t first[3] = {0, 0, 2}, count[3] = {3, 2, 1}, skip[3] = {3, 2, 3};
  uint32_t order[3] = {2, 0, 1};
  char why[128];
  if (nbody_tree_validate(NBODY_TREE_BVH, 3, is_leaf, first, count, skip, 3, order, why, sizeof why) != NBODY_OK) return 3;
  skip[1] = 1;                                                                     /* ... and with a skip link that points at itself */
  if (nbody_tree_validate(NBODY_TREE_BVH, 3, is_leaf, first, count, skip, 3, order, why, sizeof why) != NBODY_ERR_INVALID) return 4;
  if (!strstr(why, "skip")) return 5;
  /* the host builder needs no device either (bvh_tree.rs:56-96 is host code upstream) */
  float pos[8] = {1, 1, 2, 5, 7, 3, 9, 9};
  uint32_t w[4] = {1, 2, 3, 4};
  nbody_host_tree* t = NULL;
  nbody_tree_view v;
  if (nbody_host_tree_build_f32(NBODY_TREE_BVH, 4, pos, w, NULL, &t) != NBODY_OK || nbody_host_tree_info(t, &v) != NBODY_OK) return 6;
  if (v.n_nodes != 3 || v.kind != NBODY_TREE_BVH) return 7;
  nbody_host_tree_free(t);
  puts("ok");
  return 0;
}
''')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)
    hdr_cc = tmp_path / "hdr.cc"
    hdr_cc.write_text('#include "nbody_hip.h"\nint main() { return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", str(hdr_cc)], check=True)
    exe = tmp_path / "caller"
    subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lnbody_hip", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)
