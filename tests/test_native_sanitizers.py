"""CPU sanitizer runs (GPU AddressSanitizer is not available on the pool): the product's host-side tree builders
under ASan+UBSan and under TSan (they run subtrees on std::threads), the delta-stream decoder under ASan+UBSan on
damaged input, and the oracle under ASan+UBSan."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "tree_build_sanitize.cpp")


def _build_and_run(tmp_path, flags, env_extra=None, src=SRC):
    exe = str(tmp_path / os.path.basename(src)[:-4])
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-pthread", *flags, src, "-o", exe], check=True)
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK" in r.stdout and "MISMATCH" not in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr \
        and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_tree_builders_asan_ubsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


def test_tree_builders_tsan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"})


def test_delta_decoder_asan_ubsan_on_damaged_streams(tmp_path):
    """The host decoder of the delta snapshots parses bytes from outside: valid streams, then 40 000 damaged ones."""
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                   src=os.path.join(ROOT, "tests", "native", "delta_decoder_sanitize.cpp"))


def test_multi_device_barrier_and_pool_tsan(tmp_path):
    """csrc/multi_sync.hpp (the barrier in front of every collective, the one-worker-per-device pool) under TSan: votes,
    a rank failing before a barrier, and a rank failing right after the final barrier while the others still wake from it
    (a completed barrier must read `true` for everyone who took part)."""
    _build_and_run(tmp_path, ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"},
                   src=os.path.join(ROOT, "tests", "native", "multi_sync_tsan.cpp"))


def test_multi_device_barrier_and_pool_asan(tmp_path):
    _build_and_run(tmp_path, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
                   src=os.path.join(ROOT, "tests", "native", "multi_sync_tsan.cpp"))


def test_oracle_asan_ubsan(tmp_path):
    """The oracle's own ASan build, driven through ctypes in a child process with libasan preloaded."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan.so not found")
    code = r'''
import ctypes, numpy as np, sys
sys.path.insert(0, %r)
L = ctypes.CDLL(%r)
from oracle import oracle as orc
orc._LIBS["portable"] = L          # route the wrappers to the sanitized build
import nbody_simulation_amd as nb
pos, vel, w = nb.scenes.plummer(3000, seed=5)
orc.direct_accel(pos, w, targets=np.arange(50), nthreads=4)
b = orc.BVH(pos, w); b.flat(); b.walk(pos[:200], theta=0.5, nthreads=4); b.close()
q = orc.Quad(pos, w); q.flat(); q.walk(pos[:200], theta=0.5, nthreads=4); q.close()
orc.update_bvh(pos, vel, w, nsteps=3, nthreads=4); orc.update_quad(pos, vel, w, theta=0.5, nsteps=3, nthreads=4)
orc.update_direct(pos[:500], vel[:500], w[:500], nsteps=2, nthreads=4)
print("oracle-asan-ok")
''' % (ROOT, os.path.join(ROOT, "oracle", "liboracle_nbody_asan.so"))
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "oracle-asan-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
