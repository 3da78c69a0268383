import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
INPUT = os.path.join(GOLDEN, "input")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests/test_multi_device_gpu.py needs two GPUs and has therefore NEVER run (no round had such a box): it goes to the end of the session, so that
    under `-x` a first-contact failure there is reported after — not instead of — the tests that have a history."""
    first_contact = [it for it in items if it.fspath.basename == "test_multi_device_gpu.py"]
    if first_contact:
        items[:] = [it for it in items if it.fspath.basename != "test_multi_device_gpu.py"] + first_contact


def _geoms(dims, bound, fluid):
    from sphexample_amd import Fixed, Fluid, Geometry
    return [Geometry(CSVFile=os.path.join(INPUT, bound), GroupMarker=1, Type=Fixed, Dimensions=dims),
            Geometry(CSVFile=os.path.join(INPUT, fluid), GroupMarker=2, Type=Fluid, Dimensions=dims)]


def load_dam_break_2d():
    from sphexample_amd import AllocateDataStructures
    from sphexample_amd.cases import setup_dam_break_2d
    return AllocateDataStructures(_geoms(2, "DamBreak2d_Dp0.02_Bound.csv", "DamBreak2d_Dp0.02_Fluid.csv")), setup_dam_break_2d()


def load_still_wedge():
    from sphexample_amd import AllocateDataStructures, LoadMDBCNormals
    from sphexample_amd.cases import setup_still_wedge_mdbc
    p = AllocateDataStructures(_geoms(2, "StillWedge_Dp0.02_Bound.csv", "StillWedge_Dp0.02_Fluid.csv"))
    LoadMDBCNormals(p, os.path.join(INPUT, "StillWedge_Dp0.02_GhostNodes_Correct.csv"))
    return p, setup_still_wedge_mdbc()


def _load_mdbc(dims, stem_bound, stem_fluid, stem_ghost, setup):
    from sphexample_amd import AllocateDataStructures, LoadMDBCNormals
    p = AllocateDataStructures(_geoms(dims, stem_bound, stem_fluid))
    LoadMDBCNormals(p, os.path.join(INPUT, stem_ghost))
    return p, setup


def load_dam_break_2d_mdbc():
    from sphexample_amd.cases import setup_dam_break_2d_mdbc
    return _load_mdbc(2, "DamBreak2d_Dp0.02_MDBC_Bound_ThreeLayers.csv.gz", "DamBreak2d_Dp0.02_MDBC_Fluid_ThreeLayers.csv.gz",
                      "DamBreak2d_Dp0.02_MDBC_GhostNodes_ThreeLayers.csv.gz", setup_dam_break_2d_mdbc())


def load_still_wedge_middle_square():
    from sphexample_amd.cases import setup_still_wedge_middle_square_mdbc
    return _load_mdbc(2, "StillWedge_MiddleSquare_Dp0.02_Bound.csv.gz", "StillWedge_MiddleSquare_Dp0.02_Fluid.csv.gz",
                      "StillWedge_MiddleSquare_Dp0.02_GhostNodes.csv.gz", setup_still_wedge_middle_square_mdbc())


def load_duckling():
    from sphexample_amd.cases import setup_duckling_mdbc
    return _load_mdbc(3, "CaseDuckling_Dp0.01_Bound_MDBC.csv.gz", "CaseDuckling_Dp0.01_Fluid_MDBC.csv.gz",
                      "CaseDuckling_Dp0.01_GhostNodes.csv.gz", setup_duckling_mdbc())


def load_moving_square():
    """Three geometries: fixed tank (group 1), water (2), the Moving square (3) with its MotionDetails."""
    from sphexample_amd import AllocateDataStructures, Fixed, Fluid, Geometry, Moving
    from sphexample_amd.cases import moving_square_motion, setup_moving_square_2d
    g = lambda f, k, t, m=None: Geometry(CSVFile=os.path.join(INPUT, f), GroupMarker=k, Type=t, Motion=m, Dimensions=2)  # noqa: E731
    geo = [g("MovingSquare_Dp0.04_Fixed.csv.gz", 1, Fixed), g("MovingSquare_Dp0.04_Fluid.csv.gz", 2, Fluid),
           g("MovingSquare_Dp0.04_Square.csv.gz", 3, Moving, moving_square_motion())]
    p = AllocateDataStructures(geo)
    p.geometries = geo
    return p, setup_moving_square_2d(0.04)


def load_dam_break_2d_variants():
    """C1 layout with the run-time model variant of the kernel: LaminarSPS + Complex diffusion + PlanarShifting."""
    import dataclasses
    from sphexample_amd.config import ComplexDensityDiffusion, LaminarSPS, PlanarShifting
    p, s = load_dam_break_2d()
    return p, dataclasses.replace(s, SimViscosity=LaminarSPS(), SimDensityDiffusion=ComplexDensityDiffusion(),
                                  SimMetaData=dataclasses.replace(s.SimMetaData, SMode=PlanarShifting))


def load_dam_break_3d_shipped():
    from sphexample_amd import AllocateDataStructures
    from sphexample_amd.cases import setup_dam_break_3d
    return AllocateDataStructures(_geoms(3, "DamBreak3d_Dp0.02_Bound.csv.gz", "DamBreak3d_Dp0.02_Fluid.csv.gz")), setup_dam_break_3d(0.02)


@pytest.fixture(scope="session")
def dam_break_2d():
    return load_dam_break_2d()


@pytest.fixture(scope="session")
def still_wedge():
    return load_still_wedge()


@pytest.fixture(scope="session")
def dam_break_2d_mdbc():
    return load_dam_break_2d_mdbc()


@pytest.fixture(scope="session")
def still_wedge_middle_square():
    return load_still_wedge_middle_square()


@pytest.fixture(scope="session")
def dam_break_2d_variants():
    return load_dam_break_2d_variants()


@pytest.fixture(scope="session")
def moving_square():
    return load_moving_square()


@pytest.fixture(scope="session")
def duckling():
    return load_duckling()


@pytest.fixture(scope="session")
def dam_break_3d_shipped():
    return load_dam_break_3d_shipped()


def perturbed(p, seed=0, vel_scale=0.3, rho_scale=2.0, pos_scale=0.0):
    """Deterministic non-trivial state: random velocities / density offsets on top of a layout."""
    rng = np.random.default_rng(seed)
    q = p.copy()
    fluid = q.Type == 1
    q.Velocity[fluid] = rng.uniform(-vel_scale, vel_scale, size=q.Velocity[fluid].shape)
    q.Density = q.Density + rng.uniform(0, rho_scale, size=q.Density.shape)
    if pos_scale:
        q.Position[fluid] += rng.uniform(-pos_scale, pos_scale, size=q.Position[fluid].shape)
    return q


def flowing(p, seed=3, shear=1.0, base=0.4, noise=0.05, rho_scale=0.5):
    """A moving, smooth state on top of a 3-D layout: the fluid streams along +x with a shear in z (base … base + shear
    m/s) plus a little noise — fast enough that the Δx criterion (src/SPHCellList.jl:758-762) asks for a cell-list rebuild
    every few dozen steps, which the lattice at rest does not within a test's horizon."""
    rng = np.random.default_rng(seed)
    q = p.copy()
    fluid = q.Type == 1
    z = q.Position[fluid, -1]
    v = rng.uniform(-noise, noise, size=q.Velocity[fluid].shape)
    v[:, 0] += base + shear * (z - z.min()) / max(z.max() - z.min(), 1e-300)
    q.Velocity[fluid] = v
    q.Density = q.Density + rng.uniform(0, rho_scale, size=q.Density.shape)
    return q


def load_dam_break_3d_c3_flowing():
    """BASELINE config 3's lattice (dp = 0.00425, N = 1 057 738) in the `flowing` state (rank-mode workers build it themselves)."""
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    return flowing(dam_break_3d(0.00425)), setup_dam_break_3d(0.00425)
