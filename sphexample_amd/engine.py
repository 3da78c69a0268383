"""ctypes binding of libsphmi.so — the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no MI355X is visible the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from ._abi import Backend, SphmiConfig, SphmiError, make_config

_lib = None


def load_library(rebuild_if_stale: bool = True) -> C.CDLL:
    """dlopen the in-tree libsphmi.so (building it with hipcc first when the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SPHMI_LIB") or _build.LIB     # SPHMI_LIB: experiment builds (tools/sweep.py)
    if path == _build.LIB and rebuild_if_stale and _build.is_stale():
        try:
            _build.build()
        except Exception as exc:  # no hipcc on the box: fall through to whatever is on disk
            if not os.path.exists(path):
                raise RuntimeError(f"libsphmi.so is missing and cannot be built: {exc}") from exc
    if not os.path.exists(path):
        raise RuntimeError("libsphmi.so is missing: run `python -m sphexample_amd.build` (needs hipcc)")
    lib = C.CDLL(path)
    lib.sphmi_backend_info.restype = C.c_char_p
    lib.sphmi_timers.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.sphmi_force_kernel_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.sphmi_device_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def backend_info() -> str:
    return load_library().sphmi_backend_info().decode()


class Engine(Backend):
    """One simulation on one GPU behind the C ABI (`include/sphmi.h`)."""

    def __init__(self, cfg: SphmiConfig):
        super().__init__(load_library(), "sphmi_", cfg)

    def timers(self) -> dict:
        names = (C.c_char_p * 16)()
        secs = (C.c_double * 16)()
        calls = (C.c_int64 * 16)()
        n = C.c_int32()
        self._check(self._lib.sphmi_timers(self._h, 16, names, secs, calls, C.byref(n)))
        return {names[i].decode(): (secs[i], calls[i]) for i in range(n.value)}

    def force_kernel_stats(self, reset: bool = False):
        ms = C.c_double()
        n = C.c_int64()
        self._check(self._lib.sphmi_force_kernel_stats(self._h, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def make_engine(particles, setup, device_float_bytes: int = 4, device: int = 0) -> Engine:
    """Engine for a SimParticles + CaseSetup pair, with the particles uploaded."""
    host_bytes = np.dtype(particles.FloatType).itemsize
    cfg = make_config(len(particles), setup.SimConstants, setup.SimKernel, setup.SimMetaData,
                      setup.SimViscosity, setup.SimDensityDiffusion,
                      device_float_bytes=device_float_bytes, host_float_bytes=host_bytes, device=device)
    e = Engine(cfg)
    e.upload_particles(particles)
    return e


__all__ = ["Engine", "make_engine", "load_library", "backend_info", "SphmiError"]
