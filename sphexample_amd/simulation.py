"""``RunSimulation`` — host-side mirror of the reference's driver loop around the hot path.

Restates the bookkeeping of /root/reference/src/SPHCellList.jl:808-930 that surrounds
``SimulationLoop``: load mDBC normals (:827), output counter starts at 1 (:849), one engine
``advance`` per output interval (:883) with ``next_output_time`` (:687-698), stop when
``TotalTime > SimulationTime`` (:909).  VTKHDF output, logging and ParaView glue are out of scope
(SURVEY.md §2 rows 11-13); ``on_output`` receives the particles at every output time instead.
"""
from __future__ import annotations

import copy
from typing import Callable, List, Optional

import numpy as np

from ._abi import make_config
from .config import (SimulationConstants, SimulationMetaData, SPHDensityDiffusion, SPHKernelInstance,
                     SPHViscosity, next_output_time)
from .engine import Engine
from .preprocess import LoadMDBCNormals, SimParticles

# Fields of the StructArray the engine does not carry (src/PreProcess.jl:114): the reference's sort! (src/SPHCellList.jl:142)
# permutes them with everything else, so after a download they follow through the engine's own permutation
# (sphmi_download_permutation) — one gather per field, no sort on the host.
PASSIVE_FIELDS = ("ChunkID", "GravityFactor", "MotionLimiter", "BoundaryBool", "GhostNormals")


def permute_passive_fields(particles: SimParticles, prev_row, kernel_output: bool = False) -> None:
    """Row i of the downloaded arrays was row prev_row[i] at the previous call: bring the passive fields along.  Kernel /
    KernelGradient are passive too unless the handle stores them (StoreKernelOutput: they are downloaded instead)."""
    names = PASSIVE_FIELDS + (() if kernel_output else ("Kernel", "KernelGradient"))
    for k in names:
        a = getattr(particles, k)
        a[...] = np.take(a, prev_row, axis=0)          # (take: 6 ms for a million rows × 3 where a[prev_row] needs 38)


def RunSimulation(*, SimGeometry=None, SimMetaData: SimulationMetaData, SimConstants: SimulationConstants,
                  SimKernel: SPHKernelInstance, SimLogger=None, SimParticles: SimParticles,
                  SimViscosity: SPHViscosity, SimDensityDiffusion: SPHDensityDiffusion,
                  ParticleNormalsPath: Optional[str] = None,
                  on_output: Optional[Callable[[SimulationMetaData, SimParticles], None]] = None,
                  device_float_bytes: int = 0, device: int = 0, backend_factory=None,
                  async_output: bool = False) -> List[float]:
    """Same keyword signature as the reference (src/SPHCellList.jl:808-817); returns the list of
    time steps the reference collects in ``TimeSteps`` (:823,:884).  ``SimParticles`` is updated in
    place at every output time, in the engine's cell-sorted order, as the reference's is."""
    if SimMetaData.BMode.__name__ == "SimpleMDBC":
        LoadMDBCNormals(SimParticles, ParticleNormalsPath)                       # :827
    host_bytes = SimParticles.Position.dtype.itemsize
    cfg = make_config(len(SimParticles), SimConstants, SimKernel, SimMetaData, SimViscosity,
                      SimDensityDiffusion, device_float_bytes=device_float_bytes,
                      host_float_bytes=host_bytes, device=device)
    eng = (backend_factory or Engine)(cfg)
    eng.upload_particles(SimParticles)
    eng.set_motions(SimGeometry)                      # MotionDefinition, src/SPHCellList.jl:846-850
    if on_output:
        eng.pin(SimParticles)                          # the same arrays receive every output
    eng.set_clock(SimMetaData.Iteration, SimMetaData.TotalTime)
    time_steps: List[float] = []
    SimMetaData.OutputIterationCounter = 1                                       # :849
    if on_output:
        on_output(SimMetaData, SimParticles)                                     # :850
    kout = SimMetaData.KMode.__name__ == "StoreKernelOutput"

    def finish_output():
        """The fields the engine does not carry follow the sort; a StoreKernelOutput handle hands over Kernel / KernelGradient."""
        if eng._has("download_permutation"):
            permute_passive_fields(SimParticles, eng.download_permutation(), kernel_output=kout)
        if kout:
            SimParticles.Kernel[...], SimParticles.KernelGradient[...] = eng.kernel_output()

    pending = None       # async_output: metadata of the snapshot whose copies are in flight
    while True:                                                                  # :881
        prog = eng.advance(next_output_time(SimMetaData))                        # :883
        SimMetaData.Iteration = prog.iteration
        SimMetaData.CurrentTimeStep = prog.last_dt
        SimMetaData.TotalTime = prog.total_time
        SimMetaData.IndexCounter = prog.index_counter
        time_steps.append(prog.last_dt)                                          # :884
        SimMetaData.OutputIterationCounter += 1                                  # :888
        done = SimMetaData.TotalTime > SimMetaData.SimulationTime               # :909
        if on_output and async_output:
            # The copies of snapshot k run while interval k+1 is computed: the callback for k is made after the
            # NEXT advance, with the metadata captured at the snapshot.
            if pending is not None:
                eng.download_end()
                on_output(pending, SimParticles)
            eng.download_into_begin(SimParticles)
            finish_output()                    # (host-side gathers on fields that are not in flight: they overlap the copies)
            pending = copy.copy(SimMetaData)
            if done:
                eng.download_end()
                on_output(pending, SimParticles)
                break
            continue
        if on_output:
            eng.download_into(SimParticles)
            finish_output()
            on_output(SimMetaData, SimParticles)                                 # :891-894
        if done:
            if not on_output:
                eng.download_into(SimParticles)
                finish_output()
            break
    eng.unpin()
    eng.close()
    return time_steps
