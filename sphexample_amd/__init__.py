"""sphexample_amd — MI355X-native SPH neighbour + force engine behind the SPHExample API.

Host-side mirror of the names ``src/SPHExample.jl:19-62`` re-exports for the hot path; the compute
lives in ``csrc/`` (HIP, gfx950) behind the C ABI of ``include/sphmi.h``.
"""
from .config import (ArtificialViscosity, ComplexDensityDiffusion, CubicSpline, Fixed, Fluid, Geometry,  # noqa: F401
                     KernelOutputMode, Laminar, LaminarSPS, LinearDensityDiffusion, LogMode,
                     MDBCMode, MotionDetails, Moving, NoKernelOutput, NoLog, NoMDBC, NoShifting,
                     ParticleType, PlanarShifting, ShiftingMode, SimpleMDBC, SimulationConstants,
                     SimulationMetaData, SPHDensityDiffusion, SPHKernel, SPHKernelInstance,
                     SPHViscosity, StoreKernelOutput, StoreLog, WendlandC2, ZeroDensityDiffusion,
                     ZeroGravityLinearDensityDiffusion, ZeroViscosity, next_output_time)
from .preprocess import (AllocateDataStructures, LoadBoundaryNormals, LoadMDBCNormals,  # noqa: F401
                         LoadSpecificCSV, SimParticles, particles_from_arrays)

__all__ = [n for n in dir() if not n.startswith("_")]
