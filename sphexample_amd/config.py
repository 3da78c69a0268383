"""Host-side mirror of the reference's configuration types (the API the drop-in keeps intact).

Everything here is plain host data: these objects only carry parameters to the C ABI
(`include/sphmi.h`, struct ``sphmi_config``).  Names, defaults and derived-field rules follow

* ``SimulationConstants``      — /root/reference/src/SimulationConstantsConfiguration.jl:36-52
* ``SPHKernelInstance``        — /root/reference/src/SPHKernels.jl:30-72 (αD :22-23)
* ``SimulationMetaData`` + mode tags — /root/reference/src/SimulationMetaDataConfiguration.jl:12-75
* ``ParticleType``/``Geometry``/``MotionDetails`` — /root/reference/src/SimulationGeometry.jl:10-30
* viscosity / density-diffusion tag types — /root/reference/src/SPHViscosityModels.jl:16-39,
  /root/reference/src/SPHDensityDiffusionModels.jl:30,54,98,148

Julia's unicode field names are accepted as keyword aliases (``ρ₀``→``rho0``, ``α``→``alpha``,
``c₀``→``c0``, ``γ``→``gamma``, ``δᵩ``→``delta_phi``, ``ν₀``→``nu0``, ``m₀``→``m0``); the ones Python's
identifier grammar rejects (subscript digits) go through ``**{"c₀": 33.14}``.
``γ⁻¹``/``Cb⁻¹``/``h⁻¹``/``H⁻¹``/``H²``/``η²`` are spelled ``gamma_inv``/``Cb_inv``/``h_inv``/``H_inv``/``H2``/``eta2``.
"""
from __future__ import annotations

import enum
import math
import unicodedata
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

# --------------------------------------------------------------------------------------------
# tag types (dispatch tags in the reference, enum selectors at the C ABI)
# --------------------------------------------------------------------------------------------


class SPHKernel:  # src/SPHKernels.jl:10
    pass


class WendlandC2(SPHKernel):  # src/SPHKernels.jl:13
    abi_value = 0


class CubicSpline(SPHKernel):  # src/SPHKernels.jl:15-19
    abi_value = 1

    def __init__(self, eps: float = 1.0):
        self.eps = float(eps)


class SPHViscosity:  # src/SPHViscosityModels.jl:13
    abi_value: Optional[int] = None


class ZeroViscosity(SPHViscosity):
    abi_value = 0


class ArtificialViscosity(SPHViscosity):
    abi_value = 1


class Laminar(SPHViscosity):  # src/SPHViscosityModels.jl:77-87
    abi_value = 2


class LaminarSPS(SPHViscosity):  # src/SPHViscosityModels.jl:90-126
    abi_value = 3


class SPHDensityDiffusion:  # src/SPHDensityDiffusionModels.jl:22
    abi_value: Optional[int] = None


class ZeroDensityDiffusion(SPHDensityDiffusion):
    # The reference method returns vector zeros that are added to a scalar
    # (src/SPHDensityDiffusionModels.jl:44 vs src/SPHCellList.jl:295): it cannot run there.
    # The engine's value 0 simply omits the term.
    abi_value = 0


class LinearDensityDiffusion(SPHDensityDiffusion):
    abi_value = 2


class ZeroGravityLinearDensityDiffusion(SPHDensityDiffusion):  # src/SPHDensityDiffusionModels.jl:56-87
    abi_value = 1


class ComplexDensityDiffusion(SPHDensityDiffusion):  # src/SPHDensityDiffusionModels.jl:150-188
    abi_value = 3


class ShiftingMode: ...
class NoShifting(ShiftingMode): ...
class PlanarShifting(ShiftingMode): ...          # src/SPHCellList.jl:73-88,654-677
class KernelOutputMode: ...
class NoKernelOutput(KernelOutputMode): ...
class StoreKernelOutput(KernelOutputMode): ...   # src/SPHCellList.jl:106-116
class MDBCMode: ...
class NoMDBC(MDBCMode): ...
class SimpleMDBC(MDBCMode): ...
class LogMode: ...
class NoLog(LogMode): ...
class StoreLog(LogMode): ...


class ParticleType(enum.IntEnum):  # src/SimulationGeometry.jl:10-14 (UInt8 values)
    Fluid = 1
    Fixed = 2
    Moving = 3


Fluid, Fixed, Moving = ParticleType.Fluid, ParticleType.Fixed, ParticleType.Moving

# --------------------------------------------------------------------------------------------

_ALIASES = {
    "ρ0": "rho0", "m0": "m0", "α": "alpha", "c0": "c0", "γ": "gamma", "δφ": "delta_phi",
    "ν0": "nu0", "η2": "eta2", "αD": "alphaD",
}


def _norm_kwargs(kwargs: dict) -> dict:
    out = {}
    for k, v in kwargs.items():
        k2 = unicodedata.normalize("NFKC", k)
        out[_ALIASES.get(k2, k2)] = v
    return out


class SimulationConstants:
    """``SimulationConstants{T}`` — src/SimulationConstantsConfiguration.jl:36-52.

    Defaults and dependent defaults are evaluated in declaration order exactly as ``@with_kw`` does:
    ``m₀ = ρ₀·dx²`` (2-D default; the 3-D example passes ``1000·dx³``), ``c₀ = √(2g)·20``,
    ``Cb = c₀²ρ₀/γ``.
    """

    __slots__ = ("rho0", "dx", "m0", "alpha", "g", "c0", "gamma", "gamma_inv", "delta_phi", "CFL",
                 "Cb", "Cb_inv", "nu0", "BlinConstant", "SmagorinskyConstant")

    def __init__(self, **kwargs):
        kw = _norm_kwargs(kwargs)
        g = lambda name, default: float(kw.pop(name)) if name in kw else default  # noqa: E731
        self.rho0 = g("rho0", 1000.0)
        self.dx = g("dx", 0.02)
        self.m0 = g("m0", self.rho0 * self.dx ** 2)
        self.alpha = g("alpha", 0.01)
        self.g = g("g", 9.81)
        self.c0 = g("c0", math.sqrt(self.g * 2) * 20)
        self.gamma = g("gamma", 7.0)
        self.gamma_inv = g("gamma_inv", 1 / self.gamma)
        self.delta_phi = g("delta_phi", 0.1)
        self.CFL = g("CFL", 0.2)
        self.Cb = g("Cb", (self.c0 ** 2 * self.rho0) / self.gamma)
        self.Cb_inv = g("Cb_inv", 1.0 / self.Cb if self.Cb else math.inf)
        self.nu0 = g("nu0", 1e-6)
        self.BlinConstant = g("BlinConstant", 0.0066)
        self.SmagorinskyConstant = g("SmagorinskyConstant", 0.12)
        if kw:
            raise TypeError(f"unknown SimulationConstants fields: {sorted(kw)}")
        # the @assert lines of the reference struct
        for name in ("rho0", "dx", "m0", "alpha", "c0", "gamma", "gamma_inv", "delta_phi", "CFL"):
            if not getattr(self, name) > 0:
                raise AssertionError(f"{name} must be positive")
        if self.g < 0 or self.Cb < 0 or self.nu0 < 0:
            raise AssertionError("g, Cb, ν₀ must be non-negative")

    def __repr__(self):
        return "SimulationConstants(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in self.__slots__) + ")"


def _alphaD(kernel_type, dims: int, h: float) -> float:
    # src/SPHKernels.jl:22-23
    if kernel_type is WendlandC2 or isinstance(kernel_type, WendlandC2):
        if dims == 2:
            return 7 / (4 * math.pi * h ** 2)
        if dims == 3:
            return 21 / (16 * math.pi * h ** 3)
    if kernel_type is CubicSpline or isinstance(kernel_type, CubicSpline):          # :25-27
        if dims == 1:
            return 2 / (3 * h)
        if dims == 2:
            return 10 / (7 * math.pi * h ** 2)
        if dims == 3:
            return 1 / (math.pi * h ** 3)
    raise NotImplementedError("kernel / dimension combination without a normalisation constant")


class SPHKernelInstance:
    """``SPHKernelInstance{KernelType,D,T}(kernel; dx | h, k=2)`` — src/SPHKernels.jl:42-72."""

    __slots__ = ("kernel", "dims", "k", "h", "h_inv", "H", "H_inv", "H2", "alphaD", "eta2")

    def __init__(self, dims: int, kernel: SPHKernel = None, *, dx: float = None, h: float = None,
                 k: float = 2.0):
        kernel = kernel if kernel is not None else WendlandC2()
        if isinstance(kernel, type):
            kernel = kernel()
        if (dx is None) == (h is None):
            raise ValueError("Must provide exactly one of `dx` or `h`")
        h0 = k * dx if dx is not None else h
        if not h0 > 0:
            raise AssertionError("Smoothing length h must be positive")
        self.kernel = kernel
        self.dims = int(dims)
        self.k = float(k)
        self.h = float(h0)
        self.h_inv = 1.0 / self.h
        self.H = self.k * self.h
        self.H_inv = 1.0 / self.H
        self.H2 = self.H * self.H
        self.alphaD = _alphaD(kernel, self.dims, self.h)
        self.eta2 = (0.01 * self.h) ** 2


@dataclass
class MotionDetails:  # src/SimulationGeometry.jl:17-22
    Velocity: float
    StartTime: float
    Duration: float
    Direction: Sequence[float]


@dataclass
class Geometry:  # src/SimulationGeometry.jl:25-30
    CSVFile: str
    GroupMarker: int
    Type: ParticleType
    Motion: Optional[MotionDetails] = None
    Dimensions: int = 0      # the reference carries D, T as type parameters
    FloatType: str = "Float64"


@dataclass
class SimulationMetaData:
    """``SimulationMetaData{D,T,SMode,KMode,BMode,LMode}`` — src/SimulationMetaDataConfiguration.jl:28-75."""

    Dimensions: int
    FloatType: str = "Float64"
    SMode: type = NoShifting
    KMode: type = NoKernelOutput
    BMode: type = NoMDBC
    LMode: type = NoLog
    SimulationName: str = ""
    SaveLocation: str = ""
    Iteration: int = 0
    OutputEach: float = 0.02
    OutputTimes: Union[float, List[float], None] = None
    OutputIterationCounter: int = 0
    StepsTakenForLastOutput: int = 0
    CurrentTimeStep: float = 0.0
    TotalTime: float = 0.0
    SimulationTime: float = 0.0
    IndexCounter: int = 0
    VisualizeInParaview: bool = True
    ExportSingleVTKHDF: bool = True
    ExportGridCells: bool = False
    OutputVariables: List[str] = field(default_factory=lambda: [
        "ChunkID", "Kernel", "KernelGradient", "Density", "Pressure", "Velocity", "Acceleration",
        "BoundaryBool", "ID", "Type", "GroupMarker", "GhostPoints", "GhostNormals"])
    OpenLogFile: bool = True
    ChunkMultiplier: int = 1   # declared but never read by the reference

    def __post_init__(self):
        if self.OutputTimes is None:
            self.OutputTimes = self.OutputEach


def next_output_time(meta: SimulationMetaData) -> float:
    """src/SPHCellList.jl:687-698."""
    times = meta.OutputTimes
    if isinstance(times, (int, float)):
        return times * meta.OutputIterationCounter
    idx = meta.OutputIterationCounter          # 1-based in the reference
    if idx < len(times):
        return times[idx - 1]
    return meta.SimulationTime
