import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _have_gpu() -> bool:
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing); run with gpurun")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionstart(session):
    """Build the C-ABI library (hipcc cross-compiles without a GPU) and the oracle if they are missing, so that the
    suite does not depend on __graft_entry__.build() having run first.  The product itself never builds on demand."""
    import subprocess
    lib = os.path.join(ROOT, "nbody-simulation_amd", "lib", "libnbody_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "nbody-simulation_amd", "csrc"), "-j4"], check=False,
                       stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_nbody.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=False, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def nb():
    import nbody_simulation_amd
    return nbody_simulation_amd
