"""ctypes view of include/sphmi.h (structs + a thin call-marshalling base class).

``SphmiConfig`` / ``SphmiProgress`` must stay field-for-field identical to the C structs;
``tests/test_abi.py`` cross-checks the sizes against the compiled library.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .config import (NoMDBC, SimpleMDBC, SimulationConstants, SimulationMetaData, SPHDensityDiffusion,
                     SPHKernelInstance, SPHViscosity)

ABI_VERSION = 5
MAX_DEVICES = 16

OK, ERR_ARGUMENT, ERR_DEVICE, ERR_NUMERIC, ERR_DOMAIN, ERR_STATE = range(6)


class SphmiConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("abi_version", C.c_int32), ("dims", C.c_int32),
        ("host_float_bytes", C.c_int32), ("device_float_bytes", C.c_int32), ("kernel", C.c_int32),
        ("viscosity", C.c_int32), ("density_diffusion", C.c_int32), ("mdbc", C.c_int32),
        ("device", C.c_int32), ("shifting", C.c_int32), ("kernel_output", C.c_int32),
        ("n_particles", C.c_int64), ("max_cells", C.c_int64),
        ("rho0", C.c_double), ("dx", C.c_double), ("m0", C.c_double), ("alpha", C.c_double),
        ("g", C.c_double), ("c0", C.c_double), ("gamma", C.c_double), ("delta_phi", C.c_double),
        ("CFL", C.c_double), ("Cb", C.c_double), ("nu0", C.c_double),
        ("k", C.c_double), ("h", C.c_double), ("h_inv", C.c_double), ("H", C.c_double),
        ("H_inv", C.c_double), ("H2", C.c_double), ("alphaD", C.c_double), ("eta2", C.c_double),
        ("blin_constant", C.c_double), ("smagorinsky_constant", C.c_double), ("cubic_eps", C.c_double),
        ("n_devices", C.c_int32), ("slab_axis", C.c_int32), ("devices", C.c_int32 * MAX_DEVICES),
    ]


class SphmiProgress(C.Structure):
    _fields_ = [
        ("iteration", C.c_int64), ("steps_done", C.c_int64), ("n_rebuilds", C.c_int64),
        ("index_counter", C.c_int64), ("total_time", C.c_double), ("last_dt", C.c_double),
        ("delta_x", C.c_double),
    ]


class SphmiError(RuntimeError):
    def __init__(self, status: int, text: str):
        super().__init__(f"sphmi status {status}: {text}")
        self.status = status


def make_config(n_particles: int, SimConstants: SimulationConstants, SimKernel: SPHKernelInstance,
                SimMetaData: SimulationMetaData, SimViscosity: SPHViscosity,
                SimDensityDiffusion: SPHDensityDiffusion, *, device_float_bytes: int = 0,
                host_float_bytes: int = 8, device: int = 0, max_cells: int = 0) -> SphmiConfig:
    """Flatten the reference's configuration objects into the C parameter block.

    Model tags the engine does not implement raise here, which is where the Julia shim falls back
    to the stock CPU path.  device_float_bytes: 4 / 8, or 0 = the library chooses
    (`sphmi_auto_device_float_bytes`: fp32 kernels when H >= 2h without mDBC, fp64 when the kernel is cut off before it vanishes or mDBC is on)."""
    for tag, what in ((SimViscosity, "viscosity"), (SimDensityDiffusion, "density diffusion")):
        if getattr(tag, "abi_value", None) is None:
            raise NotImplementedError(f"{type(tag).__name__}: {what} model not implemented by the engine")
    if getattr(SimKernel.kernel, "abi_value", None) is None:
        raise NotImplementedError(f"{type(SimKernel.kernel).__name__}: kernel not implemented by the engine")
    c = SphmiConfig()
    c.struct_size = C.sizeof(SphmiConfig)
    c.abi_version = ABI_VERSION
    c.dims = SimMetaData.Dimensions
    c.host_float_bytes = host_float_bytes
    c.device_float_bytes = device_float_bytes
    c.kernel = SimKernel.kernel.abi_value
    c.viscosity = SimViscosity.abi_value
    c.density_diffusion = SimDensityDiffusion.abi_value
    c.mdbc = 1 if SimMetaData.BMode is SimpleMDBC else 0
    c.shifting = 1 if SimMetaData.SMode.__name__ == "PlanarShifting" else 0
    c.kernel_output = 1 if SimMetaData.KMode.__name__ == "StoreKernelOutput" else 0
    c.cubic_eps = float(getattr(SimKernel.kernel, "eps", 0.0))
    c.blin_constant = SimConstants.BlinConstant
    c.smagorinsky_constant = SimConstants.SmagorinskyConstant
    c.device = device
    c.n_particles = n_particles
    c.max_cells = max_cells
    for k in ("rho0", "dx", "m0", "alpha", "g", "c0", "gamma", "delta_phi", "CFL", "Cb", "nu0"):
        setattr(c, k, getattr(SimConstants, k))
    for k in ("k", "h", "h_inv", "H", "H_inv", "H2", "alphaD", "eta2"):
        setattr(c, k, getattr(SimKernel, k))
    return c


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Backend:
    """Call marshalling shared by the HIP engine binding and (in tests) the oracle binding.

    A backend is `lib` + symbol `prefix`; every entry point has the signature declared in
    include/sphmi.h."""

    def __init__(self, lib: C.CDLL, prefix: str, cfg: SphmiConfig, create=None):
        self._lib, self._p = lib, prefix
        self.cfg = cfg
        self.N, self.D = int(cfg.n_particles), int(cfg.dims)
        self._ft = np.float64 if cfg.host_float_bytes == 8 else np.float32
        f = self._fn
        f("last_error").restype = C.c_char_p
        f("last_error").argtypes = [C.c_void_p]
        self._h = C.c_void_p()
        f("create").argtypes = [C.POINTER(SphmiConfig), C.POINTER(C.c_void_p)]
        rc = (create or f("create"))(C.byref(cfg), C.byref(self._h))
        if rc != OK:
            raise SphmiError(rc, (f("last_error")(None) or b"").decode())
        f("destroy").argtypes = [C.c_void_p]
        f("upload").argtypes = [C.c_void_p] * 9
        f("set_clock").argtypes = [C.c_void_p, C.c_int64, C.c_double]
        f("download_kernel_output").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f("set_motion").argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_void_p]
        f("advance").argtypes = [C.c_void_p, C.c_double, C.c_int64, C.POINTER(SphmiProgress)]
        f("download").argtypes = [C.c_void_p] * 11
        f("forces_once").argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        f("unique_cells").argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _check(self, rc: int):
        if rc != OK:
            raise SphmiError(rc, (self._fn("last_error")(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data movement ----------------------------------------------------------------------
    def _f(self, a, shape):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=self._ft)
        assert a.shape == shape, (a.shape, shape)
        return a

    def upload(self, Position, Velocity, Acceleration, Density, Type, ID, GroupMarker=None,
               GhostPoints=None):
        N, D = self.N, self.D
        keep = [self._f(Position, (N, D)), self._f(Velocity, (N, D)), self._f(Acceleration, (N, D)),
                self._f(Density, (N,)), np.ascontiguousarray(Type, dtype=np.uint8),
                np.ascontiguousarray(ID, dtype=np.int64),
                None if GroupMarker is None else np.ascontiguousarray(GroupMarker, dtype=np.uint64),
                self._f(GhostPoints, (N, D))]
        self._check(self._fn("upload")(self._h, *[_ptr(a) for a in keep]))

    def upload_particles(self, p):
        self.upload(p.Position, p.Velocity, p.Acceleration, p.Density, p.Type, p.ID, p.GroupMarker,
                    p.GhostPoints if self.cfg.mdbc else None)

    def set_motion(self, group_marker: int, motion):
        """MotionDetails of one Geometry (src/SimulationGeometry.jl:17-22) → ProgressMotion of that group."""
        d = np.zeros(3)
        d[:self.D] = np.asarray(motion.Direction, dtype=np.float64)[:self.D]
        self._check(self._fn("set_motion")(self._h, int(group_marker), float(motion.Velocity), float(motion.StartTime),
                                            float(motion.Duration), d.ctypes.data_as(C.c_void_p)))

    def set_motions(self, geometries):
        """Register the Motion of every Geometry that has one (RunSimulation's MotionDefinition dict, :846-850)."""
        for g in geometries or ():
            if getattr(g, "Motion", None) is not None:
                self.set_motion(g.GroupMarker, g.Motion)

    def set_clock(self, iteration: int, total_time: float):
        self._check(self._fn("set_clock")(self._h, iteration, total_time))

    def advance(self, t_target: float, max_steps: int = -1) -> SphmiProgress:
        prog = SphmiProgress()
        self._check(self._fn("advance")(self._h, t_target, max_steps, C.byref(prog)))
        return prog

    def download(self, fields=("Position", "Velocity", "Acceleration", "Density", "Pressure", "ID",
                               "Type", "GroupMarker", "GhostPoints", "Cells"), components: Optional[int] = None) -> dict:
        """`components=3`: vector fields as n×3 (2-D vectors padded with a zero — the VTKHDF point layout the reference
        produces with to_3d!, src/ProduceHDFVTK.jl:251-325), packed that way on the device."""
        N, D = self.N, self.D
        Cv = D if components is None else int(components)
        native = Cv != D and self._has("set_output_components")
        Cd = Cv if native else D
        spec = {
            "Position": ((N, Cd), self._ft), "Velocity": ((N, Cd), self._ft),
            "Acceleration": ((N, Cd), self._ft), "Density": ((N,), self._ft),
            "Pressure": ((N,), self._ft), "ID": ((N,), np.int64), "Type": ((N,), np.uint8),
            "GroupMarker": ((N,), np.uint64), "GhostPoints": ((N, Cd), self._ft),
            "Cells": ((N, D), np.int64),
        }
        out = {k: (np.empty(*spec[k]) if k in fields else None) for k in spec}
        if native:
            self._fn("set_output_components").argtypes = [C.c_void_p, C.c_int]
            self._check(self._fn("set_output_components")(self._h, Cv))
        try:
            self._check(self._fn("download")(self._h, *[_ptr(out[k]) for k in spec]))
        finally:
            if native:
                self._check(self._fn("set_output_components")(self._h, D))
        if Cv != D and not native:          # backends without the entry point (the oracle): pad on the host
            if Cv != 3:
                raise ValueError("components must be dims or 3")
            for k in ("Position", "Velocity", "Acceleration", "GhostPoints"):
                if out[k] is not None:
                    out[k] = np.concatenate([out[k], np.zeros((N, 3 - D), dtype=out[k].dtype)], axis=1)
        return {k: v for k, v in out.items() if v is not None}

    def download_into(self, p, begin_only: bool = False) -> None:
        """Write the engine state back into the arrays of a SimParticles IN PLACE (what the reference's loop leaves in
        its StructArray).  Call pin(p) once when the same arrays receive every output."""
        order = ("Position", "Velocity", "Acceleration", "Density", "Pressure", "ID", "Type", "GroupMarker",
                 "GhostPoints", "Cells")
        want = {"Position": self._ft, "Velocity": self._ft, "Acceleration": self._ft, "Density": self._ft,
                "Pressure": self._ft, "ID": np.int64, "Type": np.uint8, "GroupMarker": np.uint64,
                "GhostPoints": self._ft, "Cells": np.int64}
        args = []
        for k in order:
            a = getattr(p, k, None)
            if not (isinstance(a, np.ndarray) and a.dtype == want[k] and a.flags.c_contiguous and len(a) == self.N):
                a = None                       # field absent or in another layout: fall back to the copying path below
            args.append(a)
        if all(a is not None for a in args):
            name = "download_begin" if begin_only and self._has("download_begin") else "download"
            self._fn(name).argtypes = [C.c_void_p] * 11
            self._check(self._fn(name)(self._h, *[_ptr(a) for a in args]))
            return
        for k, v in self.download().items():
            setattr(p, k, v)

    def download_into_begin(self, p) -> None:
        """Start an asynchronous download_into: the arrays of `p` must not be touched until download_end()."""
        self.download_into(p, begin_only=True)

    def download_end(self) -> None:
        if self._has("download_end"):
            self._fn("download_end").argtypes = [C.c_void_p]
            self._check(self._fn("download_end")(self._h))

    def pin(self, p) -> None:
        """Page-lock the field arrays of a SimParticles that will receive every output (engine backend only).  The
        arrays must stay alive until unpin() / close()."""
        if not self._has("host_register"):
            return
        self._fn("host_register").argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        self._pinned = getattr(self, "_pinned", [])
        for k in ("Position", "Velocity", "Acceleration", "Density", "Pressure", "ID", "Type", "GroupMarker", "GhostPoints", "Cells"):
            a = getattr(p, k, None)
            if isinstance(a, np.ndarray) and a.flags.c_contiguous and a.nbytes:
                self._check(self._fn("host_register")(self._h, _ptr(a), a.nbytes))
                self._pinned.append(a)

    def unpin(self) -> None:
        if not self._has("host_unregister"):
            return
        self._fn("host_unregister").argtypes = [C.c_void_p, C.c_void_p]
        for a in getattr(self, "_pinned", []):
            self._fn("host_unregister")(self._h, _ptr(a))
        self._pinned = []

    def _has(self, name: str) -> bool:
        return hasattr(self._lib, self._p + name)

    def kernel_output(self):
        """(Kernel, KernelGradient) of StoreKernelOutput runs, current order."""
        k = np.empty(self.N, dtype=self._ft)
        g = np.empty((self.N, self.D), dtype=self._ft)
        self._check(self._fn("download_kernel_output")(self._h, _ptr(k), _ptr(g)))
        return k, g

    def download_permutation(self) -> np.ndarray:
        """prev_row: row i of what download() returns now was row prev_row[i] at the previous call (at the upload for the
        first).  The reference's sort! permutes all 17 fields of the StructArray (src/SPHCellList.jl:142); this is what lets
        the caller bring along the fields the engine does not carry (`permute_passive_fields`)."""
        self._fn("download_permutation").argtypes = [C.c_void_p, C.c_void_p]
        out = np.empty(self.N, dtype=np.int64)
        self._check(self._fn("download_permutation")(self._h, _ptr(out)))
        return out

    def forces_once(self, apply_mdbc: bool = False):
        drho = np.empty(self.N, dtype=self._ft)
        acc = np.empty((self.N, self.D), dtype=self._ft)
        self._check(self._fn("forces_once")(self._h, int(apply_mdbc), _ptr(drho), _ptr(acc)))
        return drho, acc

    def unique_cells(self) -> np.ndarray:
        n = C.c_int64()
        self._check(self._fn("unique_cells")(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, self.D), dtype=np.int64)
        self._check(self._fn("unique_cells")(self._h, _ptr(out), n.value, C.byref(n)))
        return out
