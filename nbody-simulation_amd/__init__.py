"""nbody-simulation_amd — MI355X (gfx950) engine for the force + integration step of
KristinnVikarJ/nbody-simulation (reference: src/main.rs World::update, :388-425).

The product is the C-ABI shared library `lib/libnbody_hip.so` (include/nbody_hip.h) built from csrc/.
This Python package is the host-side mirror used by the tests and bench.py: a ctypes binding of that ABI
(`_capi`), `World`/`Counting` with the reference's names and call shape (`world`), the seeded initial
conditions (`scenes`) and the one-process-per-GPU sharded stepper (`sharding`).

Importable as `nbody_simulation_amd` (the repo root holds a symlink; a hyphen is not a Python identifier).
There is no CPU fallback: every compute entry point raises if the HIP library or a gfx950 device is missing.
"""
from . import _capi  # noqa: F401
from .world import Counting, World  # noqa: F401
from ._capi import DeltaDecoder  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["World", "Counting", "DeltaDecoder", "scenes", "_capi"]
