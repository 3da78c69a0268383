"""ctypes binding of libsphmi.so — the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no MI355X is visible the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from ._abi import MAX_DEVICES, OK, Backend, SphmiConfig, SphmiError, make_config

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


def _reset_library_cache() -> None:
    """Forget the loaded library so that the next load_library() follows $SPHMI_LIB again (tests of experiment builds)."""
    global _lib
    _lib = None


def backend_info() -> str:
    return load_library().sphmi_backend_info().decode()


class SphmiMultiInfo(C.Structure):
    _fields_ = [("world", C.c_int32), ("n_local", C.c_int32), ("axis", C.c_int32), ("halo_width", C.c_int32),
                ("transport", C.c_int32), ("reserved", C.c_int32), ("n_recuts", C.c_int64),
                ("cuts", C.c_int64 * MAX_DEVICES), ("n_live", C.c_int64 * MAX_DEVICES)]


def rccl_unique_id() -> bytes:
    """128 bytes for sphmi_create_rank: made on rank 0, handed to the other ranks by the launcher."""
    buf = C.create_string_buffer(128)
    rc = load_library().sphmi_rccl_unique_id(buf)
    if rc != OK:
        raise SphmiError(rc, (load_library().sphmi_last_error(None) or b"").decode())
    return buf.raw


def rccl_probe() -> None:
    """RCCL binds in this process (no id is made: `ncclGetUniqueId` starts a bootstrap root — a socket and a thread — per call)."""
    lib = load_library()
    rc = lib.sphmi_rccl_probe()
    if rc != OK:
        raise SphmiError(rc, (lib.sphmi_last_error(None) or b"").decode())


class Engine(Backend):
    """One simulation behind the C ABI (`include/sphmi.h`): on one GPU, on the GPUs of `cfg.devices` (slabs of the
    domain, all driven by this process), or — `rank=` — one slab of a run whose other slabs live in other processes."""

    def __init__(self, cfg: SphmiConfig, rank: int = None, world: int = None, unique_id: bytes = None):
        lib = load_library()
        if rank is None:
            super().__init__(lib, "sphmi_", cfg)
        else:
            lib.sphmi_create_rank.argtypes = [C.POINTER(SphmiConfig), C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)]
            create = lambda c, h: lib.sphmi_create_rank(c, rank, world, unique_id, h)  # noqa: E731
            super().__init__(lib, "sphmi_", cfg, create=create)
        self.rank_mode = rank is not None

    def multi_info(self) -> SphmiMultiInfo:
        info = SphmiMultiInfo()
        self._lib.sphmi_multi_info_get.argtypes = [C.c_void_p, C.POINTER(SphmiMultiInfo)]
        self._check(self._lib.sphmi_multi_info_get(self._h, C.byref(info)))
        return info

    @property
    def device_float_bytes(self) -> int:
        """The arithmetic this handle runs (the resolution of `device_float_bytes = 0`)."""
        n = C.c_int32()
        self._lib.sphmi_device_float_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._check(self._lib.sphmi_device_float_bytes(self._h, C.byref(n)))
        return n.value

    def owned_count(self) -> int:
        n = C.c_int64()
        self._lib.sphmi_owned_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self._check(self._lib.sphmi_owned_count(self._h, C.byref(n)))
        return n.value

    def set_cuts(self, cuts) -> None:
        """Test hook: initial slab cuts (first cell column of slabs 1 … world-1) instead of the balanced ones."""
        a = np.ascontiguousarray(cuts, dtype=np.int64)
        self._lib.sphmi_multi_set_cuts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        self._check(self._lib.sphmi_multi_set_cuts(self._h, a.ctypes.data_as(C.c_void_p), len(a)))

    def download(self, *args, **kw) -> dict:
        out = super().download(*args, **kw)
        if self.rank_mode:                 # this process's slab: the first owned_count rows
            n = self.owned_count()
            out = {k: v[:n] for k, v in out.items()}
        return out

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


def make_engine(particles, setup, device_float_bytes: int = 0, device: int = 0, devices=None, slab_axis: int = None,
                cuts=None, rank: int = None, world: int = None, unique_id: bytes = None) -> Engine:
    """Engine for a SimParticles + CaseSetup pair, with the particles uploaded.
    device_float_bytes: 4 (fp32 kernels), 8 (fp64 kernels) or 0 — the library's policy (`sphmi_auto_device_float_bytes`: fp32 where
    the kernel vanishes at its cut-off, H >= 2h, and there is no mDBC; fp64 for k < 2 — DucklingMDBC, MovingSquare2d — and for
    mDBC handles); `Engine.device_float_bytes` tells.
    devices=[0, 1, …]: one slab per listed GPU, all inside this handle (an ordinal may repeat: slabs sharing a GPU).
    rank= / world= / unique_id=: this process holds slab `rank` on GPU `device`; every process uploads the full set."""
    host_bytes = np.dtype(particles.FloatType).itemsize
    cfg = make_config(len(particles), setup.SimConstants, setup.SimKernel, setup.SimMetaData,
                      setup.SimViscosity, setup.SimDensityDiffusion,
                      device_float_bytes=device_float_bytes, host_float_bytes=host_bytes, device=device)
    if devices is not None:
        if len(devices) > MAX_DEVICES:
            raise ValueError(f"at most {MAX_DEVICES} devices per handle")
        cfg.n_devices = len(devices)
        for k, d in enumerate(devices):
            cfg.devices[k] = int(d)
    if slab_axis is not None:
        cfg.slab_axis = int(slab_axis) + 1
    e = Engine(cfg, rank=rank, world=world, unique_id=unique_id)
    if cuts is not None:
        e.set_cuts(cuts)
    if getattr(particles, "geometries", None) is not None:
        e.set_motions(particles.geometries)
    e.upload_particles(particles)
    return e


def dam_break_3d_count(dp: float):
    """(boundary, fluid) particles of the 3-D dam-break lattice at spacing dp (host formula of the library)."""
    nb, nf = C.c_int64(), C.c_int64()
    lib = load_library()
    lib.sphmi_dam_break_3d_count.argtypes = [C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    rc = lib.sphmi_dam_break_3d_count(dp, C.byref(nb), C.byref(nf))
    if rc != OK:
        raise SphmiError(rc, "sphmi_dam_break_3d_count")
    return nb.value, nf.value


def make_generated_dam_break_engine(dp: float, setup, device_float_bytes: int = 4, device: int = 0) -> Engine:
    """the 3-D dam-break lattice generated ON THE DEVICE (sphmi_generate_dam_break_3d) — no host
    arrays, no upload; the handle ends in the state make_engine(dam_break_3d(dp), setup) leaves it in."""
    nb, nf = dam_break_3d_count(dp)
    cfg = make_config(nb + nf, setup.SimConstants, setup.SimKernel, setup.SimMetaData, setup.SimViscosity,
                      setup.SimDensityDiffusion, device_float_bytes=device_float_bytes, host_float_bytes=8, device=device)
    e = Engine(cfg)
    e._lib.sphmi_generate_dam_break_3d.argtypes = [C.c_void_p, C.c_double]
    e._check(e._lib.sphmi_generate_dam_break_3d(e._h, dp))
    return e


__all__ = ["Engine", "make_engine", "make_generated_dam_break_engine", "dam_break_3d_count", "load_library", "backend_info", "rccl_unique_id", "rccl_probe", "SphmiError", "SphmiMultiInfo"]
