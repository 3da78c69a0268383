"""ctypes binding of oracle/libsphoracle.so — TEST INFRASTRUCTURE ONLY (see sph_oracle.c header).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by anything under
sphexample_amd/.  The oracle takes fp64 host arrays and computes in fp64.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from sphexample_amd._abi import Backend, SphmiConfig, make_config  # struct layout only

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsphoracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sph_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "sphmi.h")
    stale = (not os.path.exists(_LIB)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB) for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libsphoracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB


def load() -> C.CDLL:
    return C.CDLL(build())


class Oracle(Backend):
    def __init__(self, cfg: SphmiConfig, threads: int = 1):
        cfg2 = SphmiConfig.from_buffer_copy(cfg)
        cfg2.host_float_bytes = 8
        cfg2.device_float_bytes = 8
        lib = load()
        super().__init__(lib, "orc_", cfg2)
        lib.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        lib.orc_delta_t.argtypes = [C.c_void_p]
        lib.orc_delta_t.restype = C.c_double
        lib.orc_isolated_step.argtypes = [C.c_void_p]
        if threads != 1:
            self.set_threads(threads)

    def set_threads(self, n: int):
        self._lib.orc_set_threads(self._h, n)

    def delta_t(self) -> float:
        return self._lib.orc_delta_t(self._h)

    def isolated_step(self):
        self._check(self._lib.orc_isolated_step(self._h))

    def half_step_density(self) -> np.ndarray:
        """ρₙ⁺ of the last step taken (after LimitDensityAtBoundary!, src/SPHCellList.jl:778-781), in the oracle's row order."""
        out = np.empty(self.N, dtype=np.float64)
        self._lib.orc_half_step_density.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._lib.orc_half_step_density(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    @staticmethod
    def max_threads() -> int:
        return load().orc_max_threads()


def make_oracle(particles, setup, threads: int = 1) -> Oracle:
    cfg = make_config(len(particles), setup.SimConstants, setup.SimKernel, setup.SimMetaData,
                      setup.SimViscosity, setup.SimDensityDiffusion, device_float_bytes=8)
    o = Oracle(cfg, threads)
    o.upload_particles(particles)
    return o
