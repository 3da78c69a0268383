"""Case set-ups for the BASELINE configurations.

* ``dam_break_3d(dp)`` — deterministic synthetic generator of the 3-D dam-break layout.  Two of the
  reference's input blobs are absent from its checkout (``.MISSING_LARGE_BLOBS``), and the ~1 M /
  ~8 M particle configurations never existed as files, so the layout is generated.  At dp = 0.02 it
  reproduces ``input/dam_break_3d/DamBreak3d_Dp0.02_{Bound,Fluid}.csv``: the same 7 846 + 9 600
  lattice sites, the fluid block in the same order with the same IDs and densities (to the CSV's
  6 significant digits); the boundary block holds the same sites but the pillar is listed in plain
  x-major order rather than in the drawing order of the DualSPHysics export
  (tests/test_host_logic.py::test_generator_reproduces_shipped_layout checks this against the fixture copy).
* parameter presets that restate the reference's driver scripts:
  ``example/Dambreak3d.jl:8-59`` (C3/C4), ``example/StillWedgeMDBC.jl:7,30-38,60,69-71`` (C5) and the
  2-D dam-break derived from ``example/Dambreak2dMDBC.jl:7`` with the spacing the shipped files have (C1/C2).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .config import (ArtificialViscosity, Fixed, Fluid, LinearDensityDiffusion, NoMDBC, SimpleMDBC,
                     SimulationConstants, SimulationMetaData, SPHKernelInstance, WendlandC2, LaminarSPS, MotionDetails, PlanarShifting)
from .preprocess import SimParticles, particles_from_arrays


def _rnd(x: float) -> int:
    return int(math.floor(x + 0.5))


def dam_break_3d_arrays(dp: float, c0: float = 33.14, rho0: float = 1000.0, g: float = 9.81):
    """Lattice layout of the SPHERIC-style 3-D dam break with a hollow pillar.

    Returns (position[N,3], density[N], type[N] uint8, group[N], id[N]) with the boundary first
    (IDs 1..Nb, x-major / z-fastest order) and the fluid after it.
    """
    o = dp / 2
    nx, ny = _rnd(1.6 / dp) + 1, _rnd(0.66 / dp) + 1
    kwall = _rnd(0.40 / dp)
    pi0, pi1 = _rnd(0.90 / dp), _rnd(0.90 / dp) + _rnd(0.12 / dp)
    pj0, pj1 = _rnd(0.22 / dp), _rnd(0.22 / dp) + _rnd(0.14 / dp)
    kcap = _rnd(0.44 / dp)
    kmax = max(kwall, kcap)

    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(kmax + 1), indexing="ij")
    tank_perim = (ii == 0) | (ii == nx - 1) | (jj == 0) | (jj == ny - 1)
    in_pillar = (ii >= pi0) & (ii <= pi1) & (jj >= pj0) & (jj <= pj1)
    pillar_int = (ii > pi0) & (ii < pi1) & (jj > pj0) & (jj < pj1)
    pillar_perim = in_pillar & ~pillar_int
    bottom = (kk == 0) & ~pillar_int
    walls = (kk >= 1) & (kk <= kwall) & tank_perim
    shell = (kk >= 1) & (kk <= kcap - 1) & pillar_perim
    cap = (kk == kcap) & in_pillar
    # the shipped file lists the tank first and the pillar (a separate drawn object, k = 0 … cap)
    # after it; inside each object the order is x-major / z-fastest.
    tank = (bottom & ~in_pillar) | walls
    pillar = (bottom & in_pillar) | shell | cap
    bidx = np.concatenate([np.argwhere(tank), np.argwhere(pillar)])
    bpos = o + dp * bidx.astype(np.float64)

    fi, fj, fk = _rnd(0.38 / dp) + 1, _rnd(0.62 / dp) + 1, _rnd(0.28 / dp) + 1
    gi, gj, gk = np.meshgrid(np.arange(1, fi + 1), np.arange(1, fj + 1), np.arange(1, fk + 1), indexing="ij")
    fidx = np.stack([gi.ravel(), gj.ravel(), gk.ravel()], axis=1)
    fpos = o + dp * fidx.astype(np.float64)
    ztop = o + dp * fk
    B = c0 * c0 * rho0 / 7.0
    frho = rho0 * (1.0 + rho0 * g * (ztop - fpos[:, 2]) / B) ** (1.0 / 7.0)

    nb, nf = bpos.shape[0], fpos.shape[0]
    pos = np.concatenate([bpos, fpos])
    rho = np.concatenate([np.full(nb, rho0), frho])
    typ = np.concatenate([np.full(nb, int(Fixed), np.uint8), np.full(nf, int(Fluid), np.uint8)])
    grp = np.concatenate([np.full(nb, 1, np.int64), np.full(nf, 2, np.int64)])
    ids = np.arange(1, nb + nf + 1, dtype=np.int64)
    return pos, rho, typ, grp, ids


def dam_break_3d(dp: float, float_type=np.float64) -> SimParticles:
    pos, rho, typ, grp, ids = dam_break_3d_arrays(dp)
    return particles_from_arrays(3, pos, rho, typ, grp, ids, float_type)


def dam_break_3d_dp_for(n_target: float) -> float:
    """dp whose particle count is closest to n_target (count ≈ 8.4/dp³ · 0.00976 + surfaces)."""
    lo, hi = 0.001, 0.05
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if dam_break_3d_count(mid) > n_target:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def dam_break_3d_count(dp: float) -> int:
    nx, ny = _rnd(1.6 / dp) + 1, _rnd(0.66 / dp) + 1
    kwall, kcap = _rnd(0.40 / dp), _rnd(0.44 / dp)
    px, py = _rnd(0.12 / dp) + 1, _rnd(0.14 / dp) + 1
    perim = lambda a, b: 2 * (a + b) - 4  # noqa: E731
    nb = nx * ny - (px - 2) * (py - 2) + kwall * perim(nx, ny) + (kcap - 1) * perim(px, py) + px * py
    nf = (_rnd(0.38 / dp) + 1) * (_rnd(0.62 / dp) + 1) * (_rnd(0.28 / dp) + 1)
    return nb + nf


@dataclass
class CaseSetup:
    name: str
    SimConstants: SimulationConstants
    SimKernel: SPHKernelInstance
    SimMetaData: SimulationMetaData
    SimViscosity: object
    SimDensityDiffusion: object


def setup_dam_break_3d(dp: float) -> CaseSetup:
    """example/Dambreak3d.jl:8-59 with dx → dp."""
    consts = SimulationConstants(dx=dp, c0=33.14, alpha=0.1, m0=1000 * dp ** 3, CFL=0.2)
    kern = SPHKernelInstance(3, WendlandC2(), h=1 * math.sqrt(3 * dp ** 2))
    meta = SimulationMetaData(Dimensions=3, BMode=NoMDBC, SimulationName="DamBreak3D",
                              SimulationTime=1.6, OutputTimes=0.01)
    return CaseSetup("dam_break_3d", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())


def setup_dam_break_2d() -> CaseSetup:
    """C1/C2 of SURVEY.md §8d: dx = 0.02 (the spacing the shipped files have)."""
    consts = SimulationConstants(dx=0.02, c0=88.14487860902641, delta_phi=0.1, CFL=0.2, alpha=0.01)
    kern = SPHKernelInstance(2, WendlandC2(), dx=0.02)
    meta = SimulationMetaData(Dimensions=2, BMode=NoMDBC, SimulationName="DamBreak2D",
                              SimulationTime=0.05, OutputTimes=0.01)
    return CaseSetup("dam_break_2d", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())


def setup_dam_break_2d_mdbc() -> CaseSetup:
    """example/Dambreak2dMDBC.jl:7,30-36,74-81 (the script pairs dx = 0.01 with the Dp0.02 layouts; kept as is)."""
    consts = SimulationConstants(dx=0.01, c0=88.14487860902641, delta_phi=0.1, CFL=0.5, alpha=0.01)
    kern = SPHKernelInstance(2, WendlandC2(), dx=0.01)
    meta = SimulationMetaData(Dimensions=2, BMode=SimpleMDBC, SimulationName="DamBreak2D",
                              SimulationTime=2.0, OutputTimes=[0.01 * k for k in range(1, 201)])
    return CaseSetup("dam_break_2d_mdbc", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())


def setup_still_wedge_middle_square_mdbc() -> CaseSetup:
    """example/StillWedgeMiddleSquareMDBC.jl:7,30-38,50,59-61."""
    consts = SimulationConstants(dx=0.02, c0=42.48576250492629, delta_phi=0.1, CFL=0.5)
    kern = SPHKernelInstance(2, WendlandC2(), dx=0.02)
    meta = SimulationMetaData(Dimensions=2, BMode=SimpleMDBC, SimulationName="StillWedgeMiddleSquare",
                              SimulationTime=4.0, OutputTimes=0.01)
    return CaseSetup("still_wedge_middle_square_mdbc", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())


def setup_duckling_mdbc() -> CaseSetup:
    """example/DucklingMDBC.jl:7,29-42: the reference's only 3-D mDBC case (4×4 moment matrices)."""
    consts = SimulationConstants(dx=0.01, c0=23.43842998154953, delta_phi=0.1, CFL=0.2, alpha=0.02, m0=0.001)
    kern = SPHKernelInstance(3, WendlandC2(), dx=0.01, k=1.5)
    meta = SimulationMetaData(Dimensions=3, BMode=SimpleMDBC, SimulationName="CaseDuckling",
                              SimulationTime=1.0, OutputTimes=0.02)
    return CaseSetup("duckling_mdbc", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())


def setup_moving_square_2d(dx: float = 0.04) -> CaseSetup:
    """example/MovingSquare2d.jl:9-25,67,78-79 (PlanarShifting + LaminarSPS + a Moving body); dx = 0.04 is the
    resolution whose three input files the reference ships."""
    consts = SimulationConstants(dx=dx, c0=28.0, delta_phi=0.1, g=0.0, Cb=112000.0, alpha=1e-6, CFL=0.2)
    kern = SPHKernelInstance(2, WendlandC2(), dx=dx, k=math.sqrt(2))
    meta = SimulationMetaData(Dimensions=2, SMode=PlanarShifting, BMode=NoMDBC, SimulationName="MovingSquare2D",
                              SimulationTime=2.5, OutputTimes=0.01)
    return CaseSetup("moving_square_2d", consts, kern, meta, LaminarSPS(), LinearDensityDiffusion())


def moving_square_motion() -> MotionDetails:
    """example/MovingSquare2d.jl:45-50."""
    return MotionDetails(Velocity=2.8, StartTime=0.0, Duration=3.0, Direction=(1.0, 0.0))


def setup_still_wedge_mdbc() -> CaseSetup:
    """example/StillWedgeMDBC.jl:7,30-38,60,69-71."""
    consts = SimulationConstants(dx=0.02, c0=42.48576250492629, delta_phi=0.1, CFL=0.5)
    kern = SPHKernelInstance(2, WendlandC2(), dx=0.02)
    meta = SimulationMetaData(Dimensions=2, BMode=SimpleMDBC, SimulationName="StillWedge",
                              SimulationTime=4.0, OutputTimes=0.01)
    return CaseSetup("still_wedge_mdbc", consts, kern, meta, ArtificialViscosity(), LinearDensityDiffusion())
