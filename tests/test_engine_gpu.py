"""Parity of the HIP engine (through the C ABI) against the CPU oracle — needs a real MI355X.

Tolerances (north_star: "density/position error vs. CPU reference < 1e-5 relative"):
  fp64 kernels vs fp64 oracle : forces 1e-10 of the field maximum, K-step state 1e-9 relative
  fp32 kernels vs fp64 oracle : single force evaluation 2e-4 of the field maximum (fp32 cancellation
                                in Σ of O(100) pair terms), K-step density and position 1e-5 relative
"""
import os

import numpy as np
import pytest

from conftest import perturbed
from sphexample_amd import particles_from_arrays
from sphexample_amd._abi import ERR_ARGUMENT, ERR_DOMAIN, SphmiError, make_config
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d

pytestmark = pytest.mark.gpu

CASES = ["dam_break_2d", "still_wedge", "dam_break_3d_shipped"]


def by_id(st):
    order = np.argsort(st["ID"], kind="stable")
    return {k: v[order] for k, v in st.items()}


def engines(p, s, fb):
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    return make_engine(p, s, device_float_bytes=fb), make_oracle(p, s)


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("fb,tol", [(8, 1e-10), (4, 2e-4)])
def test_single_force_evaluation(case, fb, tol, request):
    p, s = request.getfixturevalue(case)
    p = perturbed(p, seed=11)
    eng, orc = engines(p, s, fb)
    d1, a1 = eng.forces_once()
    d2, a2 = orc.forces_once()
    e, o = eng.download(), orc.download()
    if fb == 8:   # identical arithmetic for the cell hash → identical stable order
        np.testing.assert_array_equal(e["ID"], o["ID"])
        np.testing.assert_array_equal(e["Cells"], o["Cells"])
        np.testing.assert_array_equal(eng.unique_cells(), orc.unique_cells())
    ie, io = np.argsort(e["ID"]), np.argsort(o["ID"])
    assert relmax(d1[ie], d2[io]) < tol
    assert relmax(a1[ie], a2[io]) < tol
    np.testing.assert_allclose(e["Pressure"][ie], o["Pressure"][io], rtol=1e-9 if fb == 8 else 3e-4,
                               atol=(1e-9 if fb == 8 else 3e-4) * np.abs(o["Pressure"]).max())
    # cell-sorted order: CartesianIndex order, last axis most significant
    c = e["Cells"]
    key = np.zeros(len(c), dtype=np.int64)
    mult = 1
    for d in range(c.shape[1]):
        key += (c[:, d] - c[:, d].min()) * mult
        mult *= int(c[:, d].max() - c[:, d].min() + 1)
    assert (np.diff(key) >= 0).all()
    same = np.diff(key) == 0
    assert (np.diff(e["ID"])[same] > 0).all()       # stable inside a cell


@pytest.mark.parametrize("case,steps", [("dam_break_2d", 40), ("dam_break_3d_shipped", 25)])
@pytest.mark.parametrize("fb,tol_rho,tol_x", [(8, 1e-9, 1e-11), (4, 1e-5, 1e-5)])
def test_k_step_parity(case, steps, fb, tol_rho, tol_x, request):
    p, s = request.getfixturevalue(case)
    eng, orc = engines(p, s, fb)
    pe = eng.advance(1e9, max_steps=steps)
    po = orc.advance(1e9, max_steps=steps)
    assert pe.iteration == po.iteration == steps
    assert pe.n_rebuilds == po.n_rebuilds
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-9 if fb == 8 else 1e-5)
    assert pe.last_dt == pytest.approx(po.last_dt, rel=1e-9 if fb == 8 else 1e-5)
    e, o = by_id(eng.download()), by_id(orc.download())
    scale_x = np.abs(o["Position"]).max()
    assert relmax(e["Density"], o["Density"]) < tol_rho
    assert np.abs(e["Position"] - o["Position"]).max() / scale_x < tol_x
    vmax = max(np.abs(o["Velocity"]).max(), 1e-12)
    assert np.abs(e["Velocity"] - o["Velocity"]).max() / vmax < (1e-8 if fb == 8 else 2e-3)
    if fb == 8:
        assert pe.index_counter == po.index_counter
        np.testing.assert_allclose(e["Acceleration"], o["Acceleration"], rtol=0,
                                   atol=1e-8 * np.abs(o["Acceleration"]).max())
        np.testing.assert_allclose(e["Pressure"], o["Pressure"], rtol=0, atol=1e-8 * np.abs(o["Pressure"]).max())


def test_output_interval_semantics(dam_break_2d):
    """Two advance calls = two SimulationLoop calls: Δx is re-armed, so each starts with a rebuild."""
    p, s = dam_break_2d
    eng, orc = engines(p, s, 8)
    for target in (2.0e-4, 5.0e-4):
        pe, po = eng.advance(target), orc.advance(target)
        assert (pe.iteration, pe.n_rebuilds, pe.steps_done) == (po.iteration, po.n_rebuilds, po.steps_done)
        assert pe.total_time == pytest.approx(po.total_time, rel=1e-12)
        assert pe.total_time > target
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-10


@pytest.mark.parametrize("fb,tol", [(8, 1e-6)])
def test_mdbc_still_wedge(still_wedge, fb, tol):
    """BASELINE config 5: StillWedge + mDBC, fp64 on the GPU, density within 1e-6 of the oracle."""
    p, s = still_wedge
    eng, orc = engines(p, s, fb)
    pe, po = eng.advance(1e9, max_steps=100), orc.advance(1e9, max_steps=100)
    assert pe.iteration == po.iteration == 100 and pe.n_rebuilds == po.n_rebuilds
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < tol
    assert np.abs(e["Position"] - o["Position"]).max() < tol * np.abs(o["Position"]).max()
    # mDBC really changed boundary densities
    bnd = o["Type"] != 1
    assert np.abs(o["Density"][bnd] - 1000.0).max() > 1e-3


def test_mdbc_single_evaluation(still_wedge):
    p, s = still_wedge
    p = perturbed(p, seed=4, vel_scale=0.0)
    for fb, tol in ((8, 1e-9), (4, 5e-4)):
        eng, orc = engines(p, s, fb)
        eng.forces_once(apply_mdbc=True)
        orc.forces_once(apply_mdbc=True)
        e, o = by_id(eng.download()), by_id(orc.download())
        assert relmax(e["Density"], o["Density"]) < tol


def test_two_particle_closed_form_on_gpu():
    """SURVEY.md appendix A through the C ABI (fp64 kernels)."""
    from test_oracle import default_2d_setup, two_particle_state
    from sphexample_amd.engine import make_engine
    eng = make_engine(two_particle_state(), default_2d_setup(), device_float_bytes=8)
    drho, acc = eng.forces_once()
    st = eng.download()
    i = int(np.where(st["ID"] == 1)[0][0]); j = 1 - i
    assert drho[i] == pytest.approx(97.80220756210991, rel=1e-11)
    assert drho[j] == pytest.approx(131.69226295405684, rel=1e-11)
    assert acc[i] == pytest.approx([16.290895561960024, 21.72119408261337], rel=1e-11)
    assert acc[j] == pytest.approx([-16.290895561960024, -21.72119408261337], rel=1e-11)


def test_isolated_particle_on_gpu():
    """Upstream KAT (test/runtests.jl:18-75) through the full step: ρ stays ρ₀, x stays, v_z falls."""
    from test_oracle import default_2d_setup
    from sphexample_amd.engine import make_engine
    s = default_2d_setup()
    p = particles_from_arrays(2, [[0.0, 0.0]], [1000.0], [1], [1], [1])
    eng = make_engine(p, s, device_float_bytes=8)
    pr = eng.advance(1e9, max_steps=200)
    st = eng.download()
    assert pr.iteration == 200
    assert abs(st["Density"][0] - 1000.0) < 1e-10
    assert st["Position"][0, 0] == 0 and st["Velocity"][0, 0] == 0 and st["Velocity"][0, 1] < 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 130])
def test_ragged_sizes_and_crowded_cell(n):
    """Tile tails (N mod 64) and many particles in ONE cell (more candidates than one chunk group)."""
    from test_oracle import default_2d_setup
    rng = np.random.default_rng(n)
    s = default_2d_setup()
    pos = rng.uniform(0.041, 0.119, size=(n, 2))           # all inside cell (1, 1)
    p = particles_from_arrays(2, pos, np.full(n, 1000.0) + rng.uniform(0, 3, n), np.ones(n), np.ones(n),
                              np.arange(1, n + 1))
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, 2))
    eng, orc = engines(p, s, 8)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    np.testing.assert_array_equal(eng.download(("ID",))["ID"], orc.download(("ID",))["ID"])
    np.testing.assert_allclose(d1, d2, rtol=0, atol=1e-10 * max(np.abs(d2).max(), 1))
    np.testing.assert_allclose(a1, a2, rtol=0, atol=1e-10 * max(np.abs(a2).max(), 1))


def test_big_crowd_exceeds_mask_slots():
    """> 32 mask slots per wave (2 300 particles within 3 cells of one row) forces mid-row drains."""
    from test_oracle import default_2d_setup
    rng = np.random.default_rng(1)
    s = default_2d_setup()
    n = 2300
    pos = np.stack([rng.uniform(0.0, 0.23, n), rng.uniform(0.041, 0.119, n)], axis=1)
    p = particles_from_arrays(2, pos, np.full(n, 1000.0), np.ones(n), np.ones(n), np.arange(1, n + 1))
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, 2))
    eng, orc = engines(p, s, 8)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    np.testing.assert_allclose(d1, d2, rtol=0, atol=1e-10 * np.abs(d2).max())
    np.testing.assert_allclose(a1, a2, rtol=0, atol=1e-10 * np.abs(a2).max())


def test_error_paths(dam_break_2d):
    from sphexample_amd.engine import Engine
    p, s = dam_break_2d
    cfg = make_config(len(p), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion)
    cfg.dims = 4
    with pytest.raises(SphmiError) as ei:
        Engine(cfg)
    assert ei.value.status == ERR_ARGUMENT
    cfg.dims = 2
    cfg.max_cells = 16
    eng = Engine(cfg)
    eng.upload_particles(p)
    with pytest.raises(SphmiError) as ei:
        eng.advance(1.0, max_steps=1)
    assert ei.value.status == ERR_DOMAIN


@pytest.mark.parametrize("slabs", [1, 2])
def test_calls_out_of_order_are_refused_not_run(dam_break_2d, slabs):
    """SPHMI_ERR_STATE (include/sphmi.h:50): every entry point that needs particles says so when there are none yet — status 5 and a text that names the
    call, never a kernel launched on empty arrays — and the handle is still good afterwards: the upload that was missing makes all of them work.  One-device
    and two-slab handles."""
    from sphexample_amd._abi import ERR_STATE
    from sphexample_amd.engine import Engine, make_engine
    p, s = dam_break_2d
    cfg = make_config(len(p), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion, device_float_bytes=8)
    if slabs > 1:
        cfg.n_devices = slabs
        for k in range(slabs):
            cfg.devices[k] = 0
    eng = Engine(cfg)
    calls = {"sphmi_advance": lambda: eng.advance(1.0, max_steps=1), "sphmi_download": lambda: eng.download(("ID",)),
             "sphmi_forces_once": lambda: eng.forces_once(), "sphmi_download_permutation": lambda: eng.download_permutation(),
             "sphmi_download_kernel_output": lambda: eng.kernel_output()}
    for name, call in calls.items():
        with pytest.raises(SphmiError) as ei:
            call()
        assert ei.value.status == ERR_STATE, (name, ei.value.status, str(ei.value))
        assert name in str(ei.value), (name, str(ei.value))
    eng.upload_particles(p)
    with pytest.raises(SphmiError) as ei:                  # uploaded now, but the handle does not store the kernel sums
        eng.kernel_output()
    assert ei.value.status == ERR_STATE and "kernel_output = STORE" in str(ei.value)
    pr = eng.advance(1e9, max_steps=3)
    ref = make_engine(p, s, device_float_bytes=8)
    pr2 = ref.advance(1e9, max_steps=3)
    assert pr.iteration == pr2.iteration == 3 and pr.total_time == pytest.approx(pr2.total_time, rel=1e-12)
    a, b = by_id(eng.download()), by_id(ref.download())
    assert relmax(a["Density"], b["Density"]) < 1e-9 and relmax(a["Position"], b["Position"]) < 1e-9
    assert sorted(eng.download_permutation().tolist()) == list(range(len(p)))


@pytest.mark.parametrize("fb,tol", [(8, 1e-10), (4, 2e-4)])
def test_eta_squared_zero(dam_break_2d, fb, tol):
    """η² = 0 is a legal SPHKernelInstance (src/SPHKernels.jl:41: η² ≥ 0): the reference's pair loop never visits i == j, the engine's accept masks hold the self
    pair — its terms must stay exact zeros, not 0·∞."""
    import dataclasses
    p, s = dam_break_2d
    p = perturbed(p, seed=5)
    p.Position += 0.513        # (Δt's viscous term divides by |x|² + η², src/TimeStepping.jl:30-37: a particle AT the origin is 0/0 in the reference too)
    import copy
    kern = copy.copy(s.SimKernel); kern.eta2 = 0.0
    s0 = dataclasses.replace(s, SimKernel=kern)
    eng, orc = engines(p, s0, fb)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    assert np.isfinite(d1).all() and np.isfinite(a1).all()
    assert relmax(d1[i1], d2[i2]) < tol and relmax(a1[i1], a2[i2]) < tol
    pe, po = eng.advance(1e9, max_steps=10), orc.advance(1e9, max_steps=10)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < (1e-9 if fb == 8 else 1e-5)


def test_a_far_flung_particle_does_not_end_the_run(dam_break_3d_shipped):
    """The reference keeps its cells in a Dict (src/SPHCellList.jl:145-157) and never refuses a domain; the engine's dense bounding grid had a default budget
    of 2^27 cells until round 5.  One fluid particle thrown far beyond the tank — a bounding grid of ≈2.9e8 cells, twice the old budget — must not end the
    run: the step goes through, matches the oracle, and the lone particle falls freely."""
    p, s = dam_break_3d_shipped
    q = p.copy()
    i = int(np.where(q.Type == 1)[0][-1])
    q.Position[i] = [22.0, 20.0, 21.0]                    # H ≈ 0.069 m: ≈ 340 × 300 × 310 cells with the padding
    eng, orc = engines(q, s, 8)
    pe, po = eng.advance(1e9, max_steps=3), orc.advance(1e9, max_steps=3)
    assert pe.iteration == po.iteration == 3 and pe.index_counter == po.index_counter
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-10 and relmax(e["Position"], o["Position"]) < 1e-12
    k = int(np.where(e["ID"] == q.ID[i])[0][0])
    assert e["Velocity"][k][2] < 0 and abs(e["Velocity"][k][0]) < 1e-12          # nobody near it: gravity only


def test_full_size_properties():
    """BASELINE config 3 size (≈1.06 M particles, fp32): size-independent properties.
    Σ m·a = 0 before gravity, sortedness + stability of the cell order, run-to-run determinism."""
    from sphexample_amd.engine import make_engine
    dp = 0.00425
    p = dam_break_3d(dp)
    s = setup_dam_break_3d(dp)
    assert 1.0e6 < len(p) < 1.1e6
    rng = np.random.default_rng(0)
    fluid = p.Type == 1
    p.Velocity[fluid] = rng.uniform(-0.5, 0.5, size=(int(fluid.sum()), 3))
    eng = make_engine(p, s, device_float_bytes=4)
    drho, acc = eng.forces_once()
    st = eng.download(("ID", "Cells", "Type"))
    assert np.isfinite(acc).all() and np.isfinite(drho).all()
    assert np.abs(acc.astype(np.float64).sum(0)).max() < 2e-5 * np.abs(acc).astype(np.float64).sum()
    c = st["Cells"]; ext = c.max(0) - c.min(0) + 1
    key = (c[:, 0] - c[:, 0].min()) + ext[0] * ((c[:, 1] - c[:, 1].min()) + ext[1] * (c[:, 2] - c[:, 2].min()))
    assert (np.diff(key) >= 0).all()
    assert (np.diff(st["ID"])[np.diff(key) == 0] > 0).all()
    eng2 = make_engine(p, s, device_float_bytes=4)
    d2, a2 = eng2.forces_once()
    np.testing.assert_array_equal(drho, d2)
    np.testing.assert_array_equal(acc, a2)
    pr = eng.advance(1e9, max_steps=5)
    pr2 = eng2.advance(1e9, max_steps=5)
    assert pr.iteration == 5 and pr.total_time == pr2.total_time
    a, b = eng.download(("Density", "Position")), eng2.download(("Density", "Position"))
    np.testing.assert_array_equal(a["Density"], b["Density"])
    np.testing.assert_array_equal(a["Position"], b["Position"])
    assert pr.last_dt == pytest.approx(0.2 * np.sqrt(3) * dp / 33.14, rel=0.2)


@pytest.mark.parametrize("dims,n,seed", [(2, 500, 1), (2, 3000, 2), (3, 2000, 3), (3, 9000, 4)])
def test_random_clouds_forces_and_steps(dims, n, seed):
    """Non-lattice input: random positions / velocities / densities / particle types (incl. Moving, whose
    GravityFactor is +1 — src/PreProcess.jl:82-84), negative coordinates (cells on both sides of 0: the
    round-half-away rule of map_floor), ragged cell populations."""
    from sphexample_amd import SimulationConstants, SimulationMetaData, SPHKernelInstance, WendlandC2, ArtificialViscosity, LinearDensityDiffusion
    from sphexample_amd.cases import CaseSetup
    rng = np.random.default_rng(seed)
    dx = 0.02
    box = (n ** (1.0 / dims)) * dx * 0.9
    pos = rng.uniform(-box / 2, box / 2, size=(n, dims))
    typ = rng.choice([1, 1, 1, 2, 3], size=n).astype(np.uint8)
    p = particles_from_arrays(dims, pos, 1000.0 + rng.uniform(-5, 15, n), typ, np.ones(n), np.arange(1, n + 1))
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, dims))
    sc = SimulationConstants(dx=dx, m0=1000 * dx ** dims, c0=40.0, alpha=0.05)
    ker = SPHKernelInstance(dims, WendlandC2(), dx=dx)
    s = CaseSetup("cloud", sc, ker, SimulationMetaData(Dimensions=dims), ArtificialViscosity(), LinearDensityDiffusion())
    eng, orc = engines(p, s, 8)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    np.testing.assert_array_equal(eng.download(("ID",))["ID"], orc.download(("ID",))["ID"])
    np.testing.assert_allclose(d1, d2, rtol=0, atol=1e-10 * np.abs(d2).max())
    np.testing.assert_allclose(a1, a2, rtol=0, atol=1e-10 * np.abs(a2).max())
    # a few steps of the (violent) cloud: same dt sequence and rebuild cadence, states to rounding
    eng, orc = engines(p, s, 8)
    pe, po = eng.advance(1e9, max_steps=6), orc.advance(1e9, max_steps=6)
    assert pe.n_rebuilds == po.n_rebuilds and pe.total_time == pytest.approx(po.total_time, rel=1e-9)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-8
    assert np.abs(e["Position"] - o["Position"]).max() < 1e-9 * box
    eng32, _ = engines(p, s, 4)
    d3, a3 = eng32.forces_once()
    i3, i2 = np.argsort(eng32.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    d2b, a2b = make_ref_forces(p, s)
    assert relmax(d3[i3], d2b) < 5e-4 and relmax(a3[i3], a2b) < 5e-4


def make_ref_forces(p, s):
    from oracle.oracle import make_oracle
    o = make_oracle(p, s)
    d, a = o.forces_once()
    i = np.argsort(o.download(("ID",))["ID"])
    return d[i], a[i]


@pytest.mark.parametrize("visc,ddt", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("fb,tol", [(8, 1e-10), (4, 5e-4)])
def test_model_switches(dam_break_2d, visc, ddt, fb, tol):
    """ZeroViscosity / ArtificialViscosity × no DDT / LinearDensityDiffusion are four compiled variants of the
    neighbour kernel (src/SPHViscosityModels.jl:43-74, src/SPHDensityDiffusionModels.jl:36-133)."""
    import dataclasses
    from sphexample_amd import ZeroViscosity, ArtificialViscosity, ZeroDensityDiffusion, LinearDensityDiffusion
    p, s = dam_break_2d
    p = perturbed(p, seed=5)
    s = dataclasses.replace(s, SimViscosity=ArtificialViscosity() if visc else ZeroViscosity(),
                            SimDensityDiffusion=LinearDensityDiffusion() if ddt else ZeroDensityDiffusion())
    eng, orc = engines(p, s, fb)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    assert relmax(d1[i1], d2[i2]) < tol and relmax(a1[i1], a2[i2]) < tol
    eng, orc = engines(p, s, fb)
    eng.advance(1e9, max_steps=10); orc.advance(1e9, max_steps=10)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < (1e-9 if fb == 8 else 1e-5)


def test_long_run_fp64_tracks_the_oracle(dam_break_3d_shipped):
    """300 steps of the shipped 3-D dam break (several rebuilds, the front is moving): same rebuild cadence,
    same clock, state to rounding.  tools/longrun_parity.py ran this to 3000 steps (83 rebuilds, impact on the
    pillar): ρ 1.6e-12, x 3.9e-12."""
    p, s = dam_break_3d_shipped
    eng, orc = engines(p, s, 8)
    pe, po = eng.advance(1e9, max_steps=300), orc.advance(1e9, max_steps=300)
    assert pe.n_rebuilds == po.n_rebuilds and pe.n_rebuilds >= 3
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-12)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-11
    assert relmax(e["Position"], o["Position"]) < 1e-11
    assert relmax(e["Velocity"], o["Velocity"]) < 1e-9


@pytest.mark.parametrize("case,steps", [("dam_break_2d_mdbc", 60), ("still_wedge_middle_square", 60), ("duckling", 25)])
def test_mdbc_examples(case, steps, request):
    """The other mDBC scripts of the reference (example/Dambreak2dMDBC.jl, StillWedgeMiddleSquareMDBC.jl and the
    3-D example/DucklingMDBC.jl with its 4×4 moment matrices — SURVEY §8 row f2): one boundary-density
    evaluation (src/SPHCellList.jl:219-266,319-365,598-622) and a run, fp64 kernels against the oracle; fp32
    kernels on the single evaluation."""
    p, s = request.getfixturevalue(case)
    q = perturbed(p, seed=8, vel_scale=0.05)
    for fb, tol in ((8, 1e-9), (4, 5e-4)):
        eng, orc = engines(q, s, fb)
        d1, a1 = eng.forces_once(apply_mdbc=True); d2, a2 = orc.forces_once(apply_mdbc=True)
        e, o = by_id(eng.download()), by_id(orc.download())
        err = np.abs(e["Density"] - o["Density"]) / np.abs(o["Density"]).max()
        if fb == 8:
            assert err.max() < tol
        else:
            # The moment matrices are fp64 in both builds, so fp32 handles differ through the rounded INPUTS only
            # (observed 5e-8 … 1.5e-7) — except at exact ties: Dambreak2dMDBC.jl pairs dx = 0.01 with a 0.02
            # lattice, H = 2·Dp, and six ghost nodes have their only fluid neighbour at r = H exactly; fp32
            # rounding puts it inside (ρ := that neighbour's density) where fp64 puts it outside (ρ unchanged).
            assert np.percentile(err, 99) < 1e-6 and (err > 1e-5).sum() <= 8
        i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
        ok = err < 1e-5
        assert relmax(a1[i1][ok], a2[i2][ok]) < max(tol, 1e-8) * 20 and relmax(d1[i1][ok], d2[i2][ok]) < max(tol, 1e-8) * 20
        bnd = o["Type"] != 1
        assert np.abs(o["Density"][bnd] - q.Density[np.argsort(q.ID)][bnd]).max() > 1e-3     # mDBC did act
    eng, orc = engines(p, s, 8)
    pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
    assert pe.iteration == po.iteration == steps and pe.n_rebuilds == po.n_rebuilds
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-10)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-9
    assert relmax(e["Position"], o["Position"]) < 1e-9


@pytest.mark.parametrize("case,steps", [("dam_break_2d_mdbc", 150), ("duckling", 40)])
@pytest.mark.parametrize("fb", [8, 4])
def test_mdbc_sixteen_lanes_per_node(case, steps, fb, request, monkeypatch):
    """k_mdbc_group (sixteen lanes per ghost node, four nodes per wave: the launch of handles with ≥ 4 096 nodes — DucklingMDBC) against
    the oracle like the one-node-per-wave kernel (src/SPHCellList.jl:219-266,319-365,598-622).  A node's sums must not depend on the
    nodes it shares a wave with — the node list is appended by atomics, its order differs from run to run: two runs, same bits.  The 2-D
    case is forced onto the kernel and runs through rebuilds on the device and on the host."""
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    q = perturbed(p, seed=11, vel_scale=0.05)
    monkeypatch.setenv("SPHMI_MDBC_GROUP", "1")
    out = []
    for _ in range(2):
        eng = make_engine(q, s, device_float_bytes=fb)
        eng.advance(1e9, max_steps=steps // 2)
        pr = eng.advance(1e9, max_steps=steps - steps // 2)           # (a second call: its opening rebuild refills the list)
        out.append((pr, by_id(eng.download())))
    (p1, e1), (p0, e0) = out
    assert p1.n_rebuilds == p0.n_rebuilds and p1.total_time == p0.total_time
    for k in ("Density", "Position", "Velocity"):
        assert np.array_equal(e1[k], e0[k]), k
    if fb == 8:
        from oracle.oracle import make_oracle
        orc = make_oracle(q, s)
        orc.advance(1e9, max_steps=steps // 2); po = orc.advance(1e9, max_steps=steps - steps // 2)
        o = by_id(orc.download())
        assert p1.n_rebuilds == po.n_rebuilds and p1.total_time == pytest.approx(po.total_time, rel=1e-10)
        assert relmax(e1["Density"], o["Density"]) < 1e-9 and relmax(e1["Position"], o["Position"]) < 1e-9


@pytest.mark.parametrize("visc,ddt", [("Laminar", "LinearDensityDiffusion"), ("LaminarSPS", "LinearDensityDiffusion"),
                                      ("ArtificialViscosity", "ZeroGravityLinearDensityDiffusion"),
                                      ("ArtificialViscosity", "ComplexDensityDiffusion"),
                                      ("LaminarSPS", "ComplexDensityDiffusion"), ("Laminar", "ZeroDensityDiffusion")])
@pytest.mark.parametrize("case", ["dam_break_2d", "dam_break_3d_shipped"])
def test_model_variants(case, visc, ddt, request):
    """the other viscosity / density-diffusion models (src/SPHViscosityModels.jl:77-126,
    src/SPHDensityDiffusionModels.jl:56-87,150-188) run through the run-time variant of the neighbour kernel."""
    import dataclasses
    import sphexample_amd.config as cfgm
    p, s = request.getfixturevalue(case)
    p = perturbed(p, seed=21)
    nu0 = 1e-3 if "Laminar" in visc else s.SimConstants.nu0        # a viscosity large enough to matter in the comparison
    consts = cfgm.SimulationConstants(**{k: getattr(s.SimConstants, k) for k in
                                         ("rho0", "dx", "m0", "alpha", "g", "c0", "gamma", "delta_phi", "CFL")}, nu0=nu0)
    s = dataclasses.replace(s, SimConstants=consts, SimViscosity=getattr(cfgm, visc)(), SimDensityDiffusion=getattr(cfgm, ddt)())
    for fb, tol in ((8, 1e-10), (4, 5e-4)):
        eng, orc = engines(p, s, fb)
        d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
        i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
        assert relmax(d1[i1], d2[i2]) < tol and relmax(a1[i1], a2[i2]) < tol
    eng, orc = engines(p, s, 8)
    pe, po = eng.advance(1e9, max_steps=12), orc.advance(1e9, max_steps=12)
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-11)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-9 and relmax(e["Velocity"], o["Velocity"]) < 1e-8


def test_model_variants_change_the_answer(dam_break_2d):
    """Guard against a variant silently falling back to the default model."""
    import dataclasses
    import sphexample_amd.config as cfgm
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    p = perturbed(p, seed=21)
    base = make_engine(p, s, device_float_bytes=8).forces_once()
    for kw in (dict(SimViscosity=cfgm.Laminar()), dict(SimViscosity=cfgm.LaminarSPS()),
               dict(SimDensityDiffusion=cfgm.ZeroGravityLinearDensityDiffusion()),
               dict(SimDensityDiffusion=cfgm.ComplexDensityDiffusion())):
        d, a = make_engine(p, dataclasses.replace(s, **kw), device_float_bytes=8).forces_once()
        assert relmax(d, base[0]) > 1e-8 or relmax(a, base[1]) > 1e-8, kw      # Complex ≈ Linear to first order: 4e-7


@pytest.mark.parametrize("fb,tol", [(8, 1e-9), (4, 2e-5)])
def test_planar_shifting(dam_break_2d, fb, tol):
    """PlanarShifting (src/SPHCellList.jl:73-88 add_shifting_terms!, :654-677 FullTimeStep) against the oracle."""
    import dataclasses
    from sphexample_amd.config import PlanarShifting
    p, s = dam_break_2d
    p = perturbed(p, seed=2, vel_scale=0.5)
    s_sh = dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, SMode=PlanarShifting))
    eng, orc = engines(p, s_sh, fb)
    eng.advance(1e9, max_steps=30); orc.advance(1e9, max_steps=30)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Position"], o["Position"]) < tol and relmax(e["Density"], o["Density"]) < tol
    plain, _ = engines(p, s, fb)
    plain.advance(1e9, max_steps=30)
    assert relmax(by_id(plain.download())["Position"], o["Position"]) > 1e-6      # shifting did move particles


@pytest.mark.parametrize("fb,tol", [(8, 1e-9), (4, 2e-2)])
def test_moving_square_example(moving_square, fb, tol):
    """example/MovingSquare2d.jl at the resolution the reference ships completely: a Moving body driven by
    ProgressMotion (src/SPHCellList.jl:575-596, called at :765 and :787), LaminarSPS, PlanarShifting."""
    p, s = moving_square
    geometries = p.geometries
    if fb == 4:
        # The lattice spacing is H/2 and k = √2 cuts the kernel where its gradient is far from zero: thousands of
        # pairs sit EXACTLY on r = H, and fp32 rounding decides them differently (4e-4 in x after 40 steps).
        # A 1e-3·dx jitter of the fluid removes the ties; and because every pair that crosses r = H while the
        # body ploughs on switches a finite force on or off (1 % density differences on single particles after
        # 10 steps), the fp32 run is a kinematics + coarse-agreement check; parity for this case is the fp64 run.
        p = perturbed(p, seed=1, vel_scale=0.0, rho_scale=0.0, pos_scale=4e-5)
    steps = 40 if fb == 8 else 10
    eng, orc = engines(p, s, fb)
    eng.set_motions(geometries); orc.set_motions(geometries)
    pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
    assert pe.n_rebuilds == po.n_rebuilds and pe.total_time == pytest.approx(po.total_time, rel=1e-9 if fb == 8 else 1e-5)
    e, o = by_id(eng.download()), by_id(orc.download())
    sq = o["Type"] == 3
    x0 = p.Position[np.argsort(p.ID)][sq]
    # fp32 handles accumulate the 80 half-step displacements at x ≈ 1.5 (6e-8 relative each)
    np.testing.assert_allclose(e["Position"][sq] - x0, [[2.8 * pe.total_time, 0.0]] * sq.sum(), atol=1e-11 if fb == 8 else 1e-5)
    np.testing.assert_allclose(e["Velocity"][sq], [[2.8, 0.0]] * sq.sum(), rtol=1e-6)
    assert relmax(e["Position"], o["Position"]) < tol and relmax(e["Density"], o["Density"]) < tol


PERF_GUARDS = os.environ.get("SPHMI_PERF_GUARDS") == "1"


@pytest.mark.parametrize("case,limit_us", [("moving_square", 55.0), ("duckling", 100.0), ("dam_break_3d_shipped", 62.0), ("dam_break_2d", 28.0)])
def test_example_step_times_stay_in_their_class(case, limit_us):
    """A guard on the step times recorded in BASELINE.md §6 (fp32: MovingSquare2d 41 µs, DucklingMDBC 76, Dambreak3d Dp0.02 47, the 2-D dam
    break 21): parity tests do not see a kernel that spills its accumulators to scratch — round 3 carried a five-fold
    slowdown of the run-time-model kernels with four and eight waves per tile (MovingSquare2d 55 → 272 µs per step) through every green suite
    until `tools/bench_examples.py` was compared with round 2's figures.
    Wall-clock limits depend on the box and its clock governor (round-5 advice): by default the limit is 2.5 × the record — a change of CLASS, which
    is what the guard is for (the structural half of it, no scratch in any kernel, is a CPU test: tests/test_bench_contract.py) — and the tight
    1.3 × only with $SPHMI_PERF_GUARDS=1 (the round's own measurement runs, tools/full_gpu_check.sh)."""
    import time
    if not PERF_GUARDS:
        limit_us = limit_us / 1.3 * 2.5
    import conftest
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + case)()
    eng = make_engine(p, s, device_float_bytes=4)
    if hasattr(p, "geometries"):
        eng.set_motions(p.geometries)
    eng.advance(1e9, max_steps=100)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        eng.advance(1e9, max_steps=300)
        best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    assert best < limit_us, f"{case}: {best:.1f} µs per step"


def test_config_3_kernel_time_stays_in_its_class():
    """The same guard on the headline: BASELINE config 3 (1 057 738 particles, fp32), the neighbour kernel's average launch — HIP events on the
    engine's own stream, what bench.py's roofline divides by — stays below 0.455 ms (round 4: 0.445, round 5: 0.436 with the f32-input matrix instructions in phase 1, 0.413-0.431 over five boxes with the f16-input one) after the clock governor has
    left its idle state."""
    from sphexample_amd.engine import make_generated_dam_break_engine
    eng = make_generated_dam_break_engine(0.00425, setup_dam_break_3d(0.00425), device_float_bytes=4)
    eng.advance(1e9, max_steps=120)
    best = 1e9
    for _ in range(3):                                   # (best of three windows: a box that hiccups once is not a slower kernel)
        eng.force_kernel_stats(reset=True)
        eng.advance(1e9, max_steps=64)
        ms, n = eng.force_kernel_stats()
        assert n > 0
        best = min(best, ms)
    limit = 0.455 if PERF_GUARDS else 0.80          # (2 × the record by default: a class guard; the tight one is opt-in, see above)
    assert best < limit, f"{best:.4f} ms per launch"


@pytest.mark.parametrize("k", [1.5, 2.0, 2.5])
@pytest.mark.parametrize("fb,tol", [(8, 1e-10), (4, 5e-4)])
def test_cutoff_other_than_2h(dam_break_2d, k, fb, tol):
    """H = k·h (src/SPHKernels.jl:57-60): with k < 2 the r² ≤ H² cut of src/SPHCellList.jl:275 removes pairs
    whose kernel gradient has not vanished yet (example/DucklingMDBC.jl uses 1.5, MovingSquare2d.jl √2)."""
    import dataclasses
    from sphexample_amd import SPHKernelInstance, WendlandC2
    p, s = dam_break_2d
    p = perturbed(p, seed=13)
    s = dataclasses.replace(s, SimKernel=SPHKernelInstance(2, WendlandC2(), dx=0.02, k=k))
    eng, orc = engines(p, s, fb)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    assert relmax(d1[i1], d2[i2]) < tol and relmax(a1[i1], a2[i2]) < tol


def test_download_into_is_in_place_and_repeatable(dam_break_2d_mdbc):
    """The output path of RunSimulation (src/SPHCellList.jl:891-894 reads the StructArray the loop mutated): the engine
    writes every field straight into the caller's arrays, page-locked on request (sphmi_host_register)."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d_mdbc
    eng = make_engine(p, s, device_float_bytes=8)
    eng.advance(1e9, max_steps=5)
    q = p.copy()
    ptrs = {k: getattr(q, k).ctypes.data for k in ("Position", "Velocity", "Density", "Pressure", "ID", "Cells", "GhostPoints")}
    for it in range(3):
        if it == 1:
            eng.pin(q)                           # page-locked from here on
        eng.download_into(q)
        assert all(getattr(q, k).ctypes.data == v for k, v in ptrs.items())
        d = eng.download()
        for k, v in d.items():
            np.testing.assert_array_equal(getattr(q, k), v, err_msg=k)
        eng.advance(1e9, max_steps=2)
    eng.unpin()
    assert (q.Density > 900).all() and np.abs(q.GhostPoints).sum() > 0 and (q.Pressure != 0).any()


def test_schedule_never_changes_results(monkeypatch):
    """The tile schedule is re-built from MEASURED work after every rebuild and the XCD shares follow measured finishing
    times — both depend on timing.  Tiles are independent and the reductions are maxima, so the particles must come out
    bit-identical with the static, estimate-based schedule and with any number of segments per XCD (the number of waves
    per tile is not a schedule: it changes the summation order)."""
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    dp = 0.0105
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    assert len(p) > 1024 * 64                               # more tiles than kWptSmall: the launches that are sampled
    out = []
    for env in ({}, {"SPHMI_RESCHED": "0", "SPHMI_XCD_FEEDBACK": "0"}, {"SPHMI_XCD_SEGS": "4"}):
        for k in ("SPHMI_RESCHED", "SPHMI_XCD_FEEDBACK", "SPHMI_WPT", "SPHMI_XCD_SEGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = make_engine(p, s, device_float_bytes=4)
        for _ in range(3):                                   # every call starts with a rebuild → a fresh measurement
            pr = e.advance(1e9, max_steps=40)
        assert pr.n_rebuilds >= 3
        out.append((pr.total_time, e.download(("Position", "Velocity", "Density", "Acceleration", "ID"))))
    for t, d in out[1:]:
        assert t == out[0][0]
        for k, v in d.items():
            np.testing.assert_array_equal(v, out[0][1][k], err_msg=k)


def test_download_in_vtkhdf_point_layout(dam_break_2d_mdbc, dam_break_3d_shipped):
    """components=3: what to_3d! (src/ProduceHDFVTK.jl:251-325) makes of the 2-D vector fields, packed on the device —
    n×3 with a zero third component; Cells stay n×dims; a 3-D handle is unchanged; the default layout is back afterwards."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d_mdbc
    eng = make_engine(p, s, device_float_bytes=4)
    eng.advance(1e9, max_steps=5)
    d2, d3 = eng.download(), eng.download(components=3)
    for k in ("Position", "Velocity", "Acceleration", "GhostPoints"):
        assert d3[k].shape == (len(p), 3)
        np.testing.assert_array_equal(d3[k][:, :2], d2[k], err_msg=k)
        assert not d3[k][:, 2].any()
    for k in ("Density", "Pressure", "ID", "Type", "GroupMarker", "Cells"):
        np.testing.assert_array_equal(d3[k], d2[k], err_msg=k)
    np.testing.assert_array_equal(eng.download()["Position"], d2["Position"])
    p3, s3 = dam_break_3d_shipped
    e3 = make_engine(p3, s3, device_float_bytes=4)
    np.testing.assert_array_equal(e3.download(components=3)["Position"], e3.download()["Position"])
    with pytest.raises(Exception):
        eng.download(components=4)


def test_run_simulation_config1_end_to_end(dam_break_2d):
    """BASELINE config 1/2 through the reference's own driver shape: RunSimulation (src/SPHCellList.jl:808-911 mirror)
    with the HIP engine against the same driver with the oracle as backend — 0.05 s of the 2-D dam break
    (≈550 steps, 5 outputs), fp64 kernels."""
    import copy
    from oracle.oracle import Oracle
    from sphexample_amd.simulation import RunSimulation
    p, s = dam_break_2d
    runs = {}
    for name, kw in (("gpu", dict(device_float_bytes=8)), ("cpu", dict(backend_factory=Oracle))):
        meta = copy.deepcopy(s.SimMetaData)
        q = p.copy()
        outs = []
        steps = RunSimulation(SimGeometry=None, SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel,
                              SimLogger=None, SimParticles=q, SimViscosity=s.SimViscosity,
                              SimDensityDiffusion=s.SimDensityDiffusion,
                              on_output=lambda m, pp: outs.append((m.OutputIterationCounter, m.TotalTime, m.Iteration)), **kw)
        runs[name] = (meta, q, outs, steps)
    (mg, qg, og, sg), (mc, qc, oc, sc) = runs["gpu"], runs["cpu"]
    assert [c for c, _, _ in og] == [c for c, _, _ in oc] and len(og) >= 6
    assert [i for _, _, i in og] == [i for _, _, i in oc] and mg.Iteration == mc.Iteration > 500
    np.testing.assert_allclose([t for _, t, _ in og], [t for _, t, _ in oc], rtol=1e-10)
    ig, ic = np.argsort(qg.ID), np.argsort(qc.ID)
    assert relmax(qg.Density[ig], qc.Density[ic]) < 1e-8 and relmax(qg.Position[ig], qc.Position[ic]) < 1e-8


@pytest.mark.parametrize("case", ["dam_break_2d", "still_wedge", "dam_break_3d_shipped"])
def test_cubic_spline_kernel_and_kernel_output(case, request):
    """SURVEY §8 row f1, last items: the CubicSpline kernel with its tensile correction (src/SPHKernels.jl:89-126; mDBC
    uses the same kernel) and StoreKernelOutput (src/SPHCellList.jl:106-116: ΣW and Σ∇W of the second pass)."""
    import dataclasses
    from sphexample_amd import CubicSpline, SPHKernelInstance, StoreKernelOutput
    p, s = request.getfixturevalue(case)
    p = perturbed(p, seed=17)
    D = s.SimMetaData.Dimensions
    kern = SPHKernelInstance(D, CubicSpline(0.2), h=s.SimKernel.h, k=s.SimKernel.k)
    s = dataclasses.replace(s, SimKernel=kern, SimMetaData=dataclasses.replace(s.SimMetaData, KMode=StoreKernelOutput))
    for fb, tol in ((8, 1e-10), (4, 5e-4)):
        eng, orc = engines(p, s, fb)
        d1, a1 = eng.forces_once(apply_mdbc=True); d2, a2 = orc.forces_once(apply_mdbc=True)
        i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
        assert relmax(d1[i1], d2[i2]) < tol and relmax(a1[i1], a2[i2]) < tol
    eng, orc = engines(p, s, 8)
    eng.advance(1e9, max_steps=8); orc.advance(1e9, max_steps=8)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-9 and relmax(e["Position"], o["Position"]) < 1e-9
    (k1, g1), (k2, g2) = eng.kernel_output(), orc.kernel_output()
    i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    assert k2.max() > 0 and relmax(k1[i1], k2[i2]) < 1e-10 and relmax(g1[i1], g2[i2]) < 1e-9


def test_kernel_output_with_the_default_kernel(dam_break_2d):
    import dataclasses
    from sphexample_amd import StoreKernelOutput
    p, s = dam_break_2d
    s = dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, KMode=StoreKernelOutput))
    eng, orc = engines(perturbed(p, seed=3), s, 8)
    eng.advance(1e9, max_steps=4); orc.advance(1e9, max_steps=4)
    (k1, g1), (k2, g2) = eng.kernel_output(), orc.kernel_output()
    i1, i2 = np.argsort(eng.download(("ID",))["ID"]), np.argsort(orc.download(("ID",))["ID"])
    assert relmax(k1[i1], k2[i2]) < 1e-10 and relmax(g1[i1], g2[i2]) < 1e-9
    # interior Shepard sum Σ Vⱼ W ≈ 1 − self term on the lattice: ΣW·V of an interior fluid particle
    assert 0.5 < np.median(k2) * s.SimConstants.m0 / 1000.0 < 1.1


def test_async_output_delivers_the_same_snapshots(dam_break_2d):
    """sphmi_download_begin / _end: the copies of an output overlap the next interval; every snapshot (state AND
    metadata) equals what the synchronous driver hands to the callback."""
    import copy
    from sphexample_amd.simulation import RunSimulation
    p, s = dam_break_2d
    got = {}
    for mode in (False, True):
        meta = copy.deepcopy(s.SimMetaData)
        meta.SimulationTime, meta.OutputTimes = 0.004, 0.001
        q = p.copy()
        snaps = []
        RunSimulation(SimGeometry=None, SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel, SimLogger=None,
                      SimParticles=q, SimViscosity=s.SimViscosity, SimDensityDiffusion=s.SimDensityDiffusion,
                      device_float_bytes=8, async_output=mode,
                      on_output=lambda m, pp: snaps.append((m.OutputIterationCounter, m.Iteration, m.TotalTime,
                                                            pp.Position.copy(), pp.Density.copy(), pp.ID.copy())))
        got[mode] = snaps
    assert len(got[True]) == len(got[False]) >= 5
    for a, b in zip(got[True][1:], got[False][1:]):          # [0] is the initial output (:850), same object state
        assert a[:3] == b[:3]
        for x, y in zip(a[3:], b[3:]):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("pinned", [True, False])
def test_download_begin_advance_end_hands_out_the_snapshot(dam_break_2d, pinned):
    """The contract of sphmi_download_begin / _end with BOTH kinds of destination: arrays the caller page-locked
    (sphmi_host_register) are written by the copy engine while the run goes on, any other array is filled from the device-side
    snapshot inside sphmi_download_end through the handle's bounce buffer (the library hands no pageable pointer to the
    runtime, profiles/HISTORY.md §4.6).  Either way the arrays hold the state of the moment of `begin`, whatever was advanced in between —
    and a second begin without an end delivers the first snapshot first."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    eng = make_engine(p, s, device_float_bytes=8)
    eng.advance(1e9, max_steps=7)
    want = eng.download()
    q = p.copy()
    if pinned:
        eng.pin(q)
    eng.download_into_begin(q)
    eng.advance(1e9, max_steps=9)                           # the state moves on while the copies fly / before they are made
    eng.download_end()
    for k in ("Position", "Velocity", "Density", "Pressure", "ID"):
        np.testing.assert_array_equal(getattr(q, k), want[k], err_msg=k)
    want2 = eng.download()
    r = p.copy()
    eng.download_into_begin(q)                              # snapshot 2 into q …
    eng.advance(1e9, max_steps=3)
    eng.download_into_begin(r)                              # … which must be complete before snapshot 3 starts
    np.testing.assert_array_equal(q.Density, want2["Density"])
    eng.download_end()
    np.testing.assert_array_equal(r.Density, eng.download()["Density"])
    assert not np.array_equal(r.Density, q.Density)
    if pinned:
        eng.unpin()


def test_output_side_against_the_oracle_backed_driver(dam_break_2d_mdbc):
    """SURVEY §8 row f3 against the ORACLE, not against the engine itself: the asynchronous, device-packed, n×3-padded
    output of the HIP engine (sphmi_download_begin / _end + sphmi_set_output_components) equals, snapshot by snapshot,
    what the same RunSimulation driver produces with the CPU oracle as backend and the padding done on the host —
    positions, densities, pressures, ghost points, cells and the UniqueCells list of the grid export."""
    import copy
    from oracle.oracle import Oracle
    from sphexample_amd.simulation import RunSimulation
    p, s = dam_break_2d_mdbc
    got = {}
    for name, kw in (("gpu", dict(device_float_bytes=8, async_output=True)), ("cpu", dict(backend_factory=Oracle))):
        meta = copy.deepcopy(s.SimMetaData)
        meta.SimulationTime, meta.OutputTimes = 0.003, 0.001
        q = p.copy()
        snaps = []
        RunSimulation(SimGeometry=None, SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel, SimLogger=None,
                      SimParticles=q, SimViscosity=s.SimViscosity, SimDensityDiffusion=s.SimDensityDiffusion,
                      on_output=lambda m, pp: snaps.append((m.OutputIterationCounter, m.Iteration, m.TotalTime, m.IndexCounter,
                                                            {k: getattr(pp, k).copy() for k in ("ID", "Position", "Density", "Pressure", "GhostPoints", "Cells")})), **kw)
        got[name] = snaps
    assert len(got["gpu"]) == len(got["cpu"]) >= 4
    for a, b in zip(got["gpu"][1:], got["cpu"][1:]):
        assert a[:2] == b[:2] and a[3] == b[3]
        assert a[2] == pytest.approx(b[2], rel=1e-12)
        np.testing.assert_array_equal(a[4]["ID"], b[4]["ID"])            # same cell-sorted order (fp64 both sides)
        np.testing.assert_array_equal(a[4]["Cells"], b[4]["Cells"])
        for k in ("Position", "Density", "Pressure", "GhostPoints"):
            assert relmax(a[4][k], b[4][k]) < 1e-9, k
    # the VTKHDF point layout (2-D vectors padded to n×3 on the device) and the grid-cell list, engine vs oracle
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    eng, orc = make_engine(p, s, device_float_bytes=8), make_oracle(p, s)
    eng.advance(1e9, max_steps=30); orc.advance(1e9, max_steps=30)
    e3, o3 = eng.download(components=3), orc.download(components=3)
    for k in ("Position", "Velocity", "Acceleration", "GhostPoints"):
        assert e3[k].shape == o3[k].shape == (len(p), 3) and not e3[k][:, 2].any()
        assert relmax(e3[k], o3[k]) < 1e-9, k
    np.testing.assert_array_equal(eng.unique_cells(), orc.unique_cells())


@pytest.mark.parametrize("dp,fb", [(0.02, 8), (0.02, 4), (0.0085, 4)])
def test_device_side_case_generator(dp, fb):
    """sphmi_generate_dam_break_3d builds the lattice on the device — same particles, order, IDs, types,
    densities as the host generator (which reproduces the reference's shipped Dp0.02 files, test_host_logic.py), and the
    run that follows is the run of the uploaded case."""
    from sphexample_amd.engine import dam_break_3d_count, make_engine, make_generated_dam_break_engine
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    nb, nf = dam_break_3d_count(dp)
    assert nb + nf == len(p) and nf == int((p.Type == 1).sum())
    up, gen = make_engine(p, s, device_float_bytes=fb), make_generated_dam_break_engine(dp, s, device_float_bytes=fb)
    a, b = up.download(), gen.download()
    for k in ("ID", "Type", "GroupMarker", "Velocity", "Acceleration"):
        np.testing.assert_array_equal(a[k], b[k])
    np.testing.assert_array_equal(a["Position"], b["Position"])
    np.testing.assert_allclose(b["Density"], a["Density"], rtol=1e-7 if fb == 4 else 1e-15, atol=0)
    np.testing.assert_allclose(b["Pressure"], a["Pressure"], rtol=0, atol=(1e-4 if fb == 4 else 1e-9) * np.abs(a["Pressure"]).max())
    pa, pb = up.advance(1e9, max_steps=12), gen.advance(1e9, max_steps=12)
    assert (pa.iteration, pa.n_rebuilds) == (pb.iteration, pb.n_rebuilds)
    assert pb.total_time == pytest.approx(pa.total_time, rel=1e-6 if fb == 4 else 1e-12)
    a, b = by_id(up.download()), by_id(gen.download())
    assert relmax(b["Density"], a["Density"]) < (1e-6 if fb == 4 else 1e-12)
    assert relmax(b["Position"], a["Position"]) < (1e-6 if fb == 4 else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", ["1", "0"])
def test_control_inside_the_predictor_keeps_the_step_sequence(dam_break_2d, fuse, monkeypatch):
    """Plain handles take the step control inside the predictor (two control blocks / two sets of reduction slots that flip
    per queued step; cancelled steps of a batch consume nothing).  Short calls of odd and even length, by step count and by
    time, each starting with a rebuild: Δt, the loop counters and the state must follow the oracle call by call — with the
    one-thread control launch (SPHMI_FUSE_CTRL=0) and without."""
    monkeypatch.setenv("SPHMI_FUSE_CTRL", fuse)
    p, s = dam_break_2d
    eng, orc = engines(p, s, 8)
    now = 0.0
    for k, n in enumerate([1, 2, 3, 1, 5, 8, 2, 17, 1, 4]):
        if k % 3 == 2:                         # every third call ends by TIME: its last control returns with unconsumed maxima
            pe, po = eng.advance(now + 2.7e-4 * n), orc.advance(now + 2.7e-4 * n)
        else:
            pe, po = eng.advance(1e9, max_steps=n), orc.advance(1e9, max_steps=n)
        assert (pe.iteration, pe.steps_done, pe.n_rebuilds) == (po.iteration, po.steps_done, po.n_rebuilds)
        assert pe.last_dt == pytest.approx(po.last_dt, rel=1e-10)
        assert pe.total_time == pytest.approx(po.total_time, rel=1e-12)
        now = po.total_time
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-9
    assert np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("case,dt_unit", [("still_wedge", 3.2e-4), ("dam_break_2d_mdbc", 9e-5)])
def test_control_inside_the_mdbc_kernel_keeps_the_step_sequence(case, dt_unit, fuse, request, monkeypatch):
    """mDBC handles take the step control inside k_mdbc, the first kernel of their step (MdbcParams::ctl_in): its bad-density
    flag goes to the slot set the step's corrector fills, which the PREVIOUS step's predictor has cleared.  The same short
    calls as above — odd and even lengths, by step count and by time, every call opening with a rebuild and a cancelled
    control — against the oracle call by call, with the fusion (default) and with the one-thread control launch
    (SPHMI_FUSE_MDBC=0)."""
    monkeypatch.setenv("SPHMI_FUSE_MDBC", fuse)
    p, s = request.getfixturevalue(case)
    eng, orc = engines(p, s, 8)
    now = 0.0
    for k, n in enumerate([1, 2, 3, 1, 5, 8, 2, 17, 1, 4]):
        if k % 3 == 2:
            pe, po = eng.advance(now + dt_unit * n), orc.advance(now + dt_unit * n)
        else:
            pe, po = eng.advance(1e9, max_steps=n), orc.advance(1e9, max_steps=n)
        assert (pe.iteration, pe.steps_done, pe.n_rebuilds) == (po.iteration, po.steps_done, po.n_rebuilds)
        assert pe.last_dt == pytest.approx(po.last_dt, rel=1e-9)
        assert pe.total_time == pytest.approx(po.total_time, rel=1e-11)
        now = po.total_time
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-8
    assert np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("wpt", ["1", "2", "4", "8"])
@pytest.mark.parametrize("case,steps", [("dam_break_3d_shipped", 12), ("dam_break_2d", 20)])
def test_every_waves_per_tile_variant_matches_the_oracle(case, steps, wpt, request, monkeypatch):
    """The number of waves per tile follows the tile count (8 / 4 / 2 / 1; two-wave tiles of 3-D handles go out in pairs, one-wave
    tiles four per block): here every variant is forced on the same small cases — partial last blocks, tiles of ghosts of the
    pairing included — and has to track the fp64 oracle like the default choice."""
    monkeypatch.setenv("SPHMI_WPT", wpt)
    p, s = request.getfixturevalue(case)
    for fb, tol in ((8, 1e-9), (4, 1e-5)):
        eng, orc = engines(p, s, fb)
        pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
        assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
        assert pe.last_dt == pytest.approx(po.last_dt, rel=1e-9 if fb == 8 else 1e-5)
        e, o = by_id(eng.download()), by_id(orc.download())
        assert relmax(e["Density"], o["Density"]) < tol
        assert np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max() < tol


@pytest.mark.gpu
def test_mid_size_launch_matches_the_oracle():
    """≈85 k particles = 1.3 k tiles: the launches that fit the chip at once (two waves per tile, tiles in pairs, sixteen cost
    classes, measured-work order) against the oracle on the generated lattice — no shipped layout has this size."""
    from oracle.oracle import make_oracle
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    dp = 0.0105
    p, s = perturbed(dam_break_3d(dp), seed=5), setup_dam_break_3d(dp)
    orc = make_oracle(p, s)
    po = orc.advance(1e9, max_steps=6)
    o = by_id(orc.download())
    for fb, tol in ((8, 1e-9), (4, 1e-5)):
        eng = make_engine(p, s, device_float_bytes=fb)
        pe = eng.advance(1e9, max_steps=6)
        assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
        e = by_id(eng.download())
        assert relmax(e["Density"], o["Density"]) < tol
        assert np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max() < tol


@pytest.mark.parametrize("case,fb", [("dam_break_3d_shipped", 4), ("dam_break_2d", 8), ("dam_break_2d_mdbc", 8)])
def test_identity_rebuilds_change_nothing(case, fb, request, monkeypatch):
    """Every sphmi_advance opens with UpdateNeighbors! (Δx re-armed, src/SPHCellList.jl:739).  When no particle has left the cell
    it was sorted into, the reference's stable sort (:142) is the identity and the engine stops after the test that says so
    (k_cell_bbox compares every particle's cell with the key of the last sort).  Step-by-step driving from rest — every call a
    rebuild, almost all of them identities — must give bit for bit what the always-sorting engine gives, the same counters
    as the oracle, and must actually take the short cut."""
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    out = {}
    monkeypatch.setenv("SPHMI_DEVICE_REBUILD", "0")     # (the short cut belongs to the host-side rebuild; handles as small as these rebuild on the device by default)
    for flag in ("1", "0"):
        monkeypatch.setenv("SPHMI_SAME_CELLS", flag)
        eng = make_engine(p, s, device_float_bytes=fb)
        prog = [eng.advance(1e9, max_steps=2) for _ in range(12)]
        out[flag] = (eng.download(), [(q.iteration, q.n_rebuilds, q.index_counter, q.total_time, q.last_dt) for q in prog],
                     eng.timers()["02b UpdateNeighbors calls that were the identity (no sort)"][1], eng.unique_cells())
    for k, v in out["1"][0].items():
        np.testing.assert_array_equal(v, out["0"][0][k], err_msg=k)
    assert out["1"][1] == out["0"][1]
    np.testing.assert_array_equal(out["1"][3], out["0"][3])
    # (the 3-D column at rest keeps every particle in its cell over these 24 steps; the collapsing 2-D columns lose one now and then)
    assert out["0"][2] == 0 and out["1"][2] >= (6 if case == "dam_break_3d_shipped" else 0), (out["0"][2], out["1"][2])
    orc = engines(p, s, fb)[1]
    po = [orc.advance(1e9, max_steps=2) for _ in range(12)]
    assert [(q.iteration, q.n_rebuilds, q.index_counter) for q in po] == [t[:3] for t in out["1"][1]]
    e, o = by_id(out["1"][0]), by_id(orc.download())
    np.testing.assert_array_equal(out["1"][0]["ID"], orc.download(("ID",))["ID"]) if fb == 8 else None
    assert relmax(e["Density"], o["Density"]) < (1e-9 if fb == 8 else 1e-5)


def test_ghost_points_follow_the_sort_without_mdbc(dam_break_2d):
    """GhostPoints is one of the 17 columns the reference's sort! permutes (src/SPHCellList.jl:142) whether or not mDBC reads
    it.  A NoMDBC handle that was given ghost points hands them back row-aligned after any number of rebuilds (the column used
    to travel for mDBC handles only: a download after an odd number of rebuilds showed the other, never-written buffer)."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    q = perturbed(p, seed=4, vel_scale=3.0)
    q.GhostPoints[:] = np.stack([q.ID * 1.5, -q.ID * 0.25], axis=1)
    eng = make_engine(q, s, device_float_bytes=8)
    eng.upload(q.Position, q.Velocity, q.Acceleration, q.Density, q.Type, q.ID, q.GroupMarker, q.GhostPoints)
    for calls in range(1, 6):
        pr = eng.advance(1e9, max_steps=9)
        d = eng.download(("ID", "GhostPoints"))
        np.testing.assert_array_equal(d["GhostPoints"], np.stack([d["ID"] * 1.5, -d["ID"] * 0.25], axis=1), err_msg=f"after {pr.n_rebuilds} rebuilds")
    assert pr.n_rebuilds >= 5


# ---- device-side rebuilds (round 4): small handles rebuild their cell list without asking the host ---------------------------------
def _counter(eng, prefix):
    return next(v[1] for k, v in eng.timers().items() if k.startswith(prefix))


@pytest.mark.parametrize("case,fb,vel", [("dam_break_2d", 8, 3.0), ("dam_break_2d", 4, 3.0), ("dam_break_3d_shipped", 8, 3.0), ("dam_break_3d_shipped", 4, 3.0),
                                         ("dam_break_2d_mdbc", 8, 2.0), ("moving_square", 8, 0.0), ("duckling", 4, 1.0)])
def test_device_side_rebuild_is_the_host_side_rebuild(case, fb, vel, request, monkeypatch):
    """UpdateNeighbors! (src/SPHCellList.jl:138-163) on the device's own say-so — the grid of the last host-side rebuild plus two
    cell layers, k_cell_count → k_scan_single → k_scatter → k_rankfix → k_permute → k_tile_schedule_small, the force launches on an
    upper-bound grid until the run table has been seen — against the host path ($SPHMI_DEVICE_REBUILD=0: bounding box, exact grid,
    three synchronisations): BIT FOR BIT the same state, order, cells and loop counters, call after call, and the oracle's
    rebuild steps and IndexCounter."""
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    if vel:
        p = perturbed(p, seed=3, vel_scale=vel)               # fast enough for several Δx-triggered rebuilds in the window
    dev = make_engine(p, s, device_float_bytes=fb)
    monkeypatch.setenv("SPHMI_DEVICE_REBUILD", "0")
    host = make_engine(p, s, device_float_bytes=fb)
    monkeypatch.delenv("SPHMI_DEVICE_REBUILD")
    orc = make_oracle(p, s)
    for e in (dev, host):
        if hasattr(p, "geometries"):
            e.set_motions(p.geometries)
    if hasattr(p, "geometries"):
        orc.set_motions(p.geometries)
    for steps in (1, 40, 37, 2, 60):
        pd, ph, po = dev.advance(1e9, max_steps=steps), host.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
        assert (pd.iteration, pd.steps_done, pd.n_rebuilds, pd.index_counter, pd.total_time, pd.last_dt) == \
               (ph.iteration, ph.steps_done, ph.n_rebuilds, ph.index_counter, ph.total_time, ph.last_dt)
        assert (pd.iteration, pd.n_rebuilds) == (po.iteration, po.n_rebuilds)
        if fb == 8:
            assert pd.index_counter == po.index_counter
        d, h = dev.download(), host.download()
        for k in d:
            np.testing.assert_array_equal(d[k], h[k], err_msg=k)
        np.testing.assert_array_equal(dev.unique_cells(), host.unique_cells())
    assert pd.n_rebuilds >= 5
    assert _counter(dev, "02c") >= pd.n_rebuilds - 1 - _counter(dev, "02d") and _counter(host, "02c") == 0
    for e in (dev, host, orc):
        e.close()


@pytest.mark.parametrize("dims,fb", [(2, 8), (3, 8), (3, 4)])
def test_a_particle_leaving_the_sticky_grid_sends_the_rebuild_to_the_host(dims, fb, monkeypatch):
    """A free blob of fluid flying apart: the bounding box grows with every rebuild interval, so a device-side rebuild sooner or later
    finds a particle outside the grid it was given.  That rebuild must change NOTHING (it copies instead of permuting), the control
    must cancel every step queued behind it (error 3, never seen by the caller), and the host-side rebuild that follows must leave
    exactly what a handle that always rebuilds on the host has: same order (the in-cell order is a history, Q4), same state."""
    from oracle.oracle import make_oracle
    from sphexample_amd import (ArtificialViscosity, LinearDensityDiffusion, SimulationConstants, SimulationMetaData, SPHKernelInstance,
                                WendlandC2, particles_from_arrays)
    from sphexample_amd.cases import CaseSetup
    from sphexample_amd.engine import make_engine
    rng = np.random.default_rng(7)
    dx = 0.02
    m = 18 if dims == 2 else 9
    g = np.stack(np.meshgrid(*[np.arange(m)] * dims, indexing="ij"), -1).reshape(-1, dims) * dx + rng.uniform(-1e-3, 1e-3, size=(m ** dims, dims))
    n = len(g)
    p = particles_from_arrays(dims, g, np.full(n, 1000.0), np.ones(n, np.uint8), np.ones(n, np.int64), np.arange(n) + 1)
    c = g - g.mean(0)
    p.Velocity[:] = (60.0 if dims == 3 else 150.0) * c / np.abs(c).max()      # outwards: a cell of H = 0.08 every ≈15 (3-D) / ≈6 (2-D) steps
    sc = SimulationConstants(dx=dx, m0=1000 * dx ** dims, c0=80.0, alpha=0.01, g=0.0, CFL=0.2)
    s = CaseSetup("blob", sc, SPHKernelInstance(dims, WendlandC2(), dx=dx, k=2.0), SimulationMetaData(Dimensions=dims), ArtificialViscosity(), LinearDensityDiffusion())
    dev = make_engine(p, s, device_float_bytes=fb)
    monkeypatch.setenv("SPHMI_DEVICE_REBUILD", "0")
    host = make_engine(p, s, device_float_bytes=fb)
    monkeypatch.delenv("SPHMI_DEVICE_REBUILD")
    orc = make_oracle(p, s)
    for steps in (60, 1, 90):
        pd, ph, po = dev.advance(1e9, max_steps=steps), host.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
        assert (pd.iteration, pd.n_rebuilds, pd.index_counter, pd.total_time) == (ph.iteration, ph.n_rebuilds, ph.index_counter, ph.total_time)
        assert (pd.iteration, pd.n_rebuilds) == (po.iteration, po.n_rebuilds)
        d, h = dev.download(), host.download()
        for k in d:
            np.testing.assert_array_equal(d[k], h[k], err_msg=k)
    assert _counter(dev, "02d") >= 1 and _counter(dev, "02c") >= 2, dev.timers()
    if fb == 8:
        o = orc.download()
        np.testing.assert_array_equal(d["ID"], o["ID"])
        assert relmax(d["Density"], o["Density"]) < 1e-9 and np.abs(d["Position"] - o["Position"]).max() < 1e-9 * np.abs(o["Position"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("wpt", ["2", "4", "8"])
def test_half_tiles_with_several_waves_per_half_on_crowds(wpt, dam_break_3d_shipped, dam_break_2d_mdbc, monkeypatch):
    """Every kernel of two, four or eight waves per tile serves half tiles (32 targets, two lanes each; profiles/HISTORY.md §4.8); with four / eight
    waves the two / four waves of a half deal its chunks alternately and hand their sums to the first through LDS.  Each is forced
    here on crowds whose rows hold dozens of chunks (so that every wave of a half gets several, and the queues drain in bursts) and
    over K steps of a dense and of a sparse layout (Dambreak3d Dp0.02; Dambreak2dMDBC, four particles per cell), against the oracle."""
    from test_oracle import default_2d_setup
    monkeypatch.setenv("SPHMI_WPT", wpt)
    rng = np.random.default_rng(7)
    s2 = default_2d_setup()
    n = 2300                                                # 3 cells of one row: 36 chunks per row, 108 per tile
    pos = np.stack([rng.uniform(0.0, 0.23, n), rng.uniform(0.041, 0.119, n)], axis=1)
    p = particles_from_arrays(2, pos, np.full(n, 1000.0) + rng.uniform(0, 3, n), np.ones(n), np.ones(n), np.arange(1, n + 1))
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, 2))
    for fb, tol in ((8, 1e-10), (4, 2e-4)):
        eng, orc = engines(p, s2, fb)
        d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
        ie, io = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
        np.testing.assert_allclose(d1[ie], d2[io], rtol=0, atol=tol * np.abs(d2).max())
        np.testing.assert_allclose(a1[ie], a2[io], rtol=0, atol=tol * np.abs(a2).max())
    dp = 0.02
    s3 = setup_dam_break_3d(dp)
    n = 3000                                                # 3-D crowd: two cells of one row, 47 chunks per row
    H = s3.SimKernel.H
    pos = np.stack([rng.uniform(0.6 * H, 2.4 * H, n), rng.uniform(0.6 * H, 1.4 * H, n), rng.uniform(0.6 * H, 1.4 * H, n)], axis=1)
    p = particles_from_arrays(3, pos, np.full(n, 1000.0) + rng.uniform(0, 3, n), np.ones(n), np.ones(n), np.arange(1, n + 1))
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, 3))
    eng, orc = engines(p, s3, 8)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    ie, io = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
    np.testing.assert_allclose(d1[ie], d2[io], rtol=0, atol=1e-10 * np.abs(d2).max())
    np.testing.assert_allclose(a1[ie], a2[io], rtol=0, atol=1e-10 * np.abs(a2).max())
    for (p, s), steps in ((dam_break_3d_shipped, 12), (dam_break_2d_mdbc, 20)):
        for fb, tol in ((8, 1e-9), (4, 1e-5)):
            eng, orc = engines(p, s, fb)
            pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
            assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
            e, o = by_id(eng.download()), by_id(orc.download())
            assert relmax(e["Density"], o["Density"]) < tol
            assert np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("wpt", ["2", "4", "8"])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 63, 64, 65, 97, 130, 1000])
def test_half_tiles_on_ragged_sizes(n, wpt, monkeypatch):
    """Half tiles of 32 targets with two lanes per target (profiles/HISTORY.md §4.8), one, two or four waves per half: forced here on particle counts
    whose last tile leaves the waves of the second half with no target, one target, or a partial set, in 2-D and 3-D, fp32 and fp64 —
    forces of one evaluation and the state after three steps against the oracle."""
    from test_oracle import default_2d_setup
    monkeypatch.setenv("SPHMI_WPT", wpt)
    rng = np.random.default_rng(100 + n)
    for dims, s in ((2, default_2d_setup()), (3, setup_dam_break_3d(0.02))):
        H = s.SimKernel.H
        side = max(1.0, (n / 12.0) ** (1.0 / dims))                      # ≈12 particles per cell
        pos = rng.uniform(0.6 * H, (0.6 + side) * H, size=(n, dims))
        ty = np.where(rng.random(n) < 0.8, 1, 2)
        p = particles_from_arrays(dims, pos, np.full(n, 1000.0) + rng.uniform(0, 3, n), ty, ty, np.arange(1, n + 1))
        p.Velocity[:] = rng.uniform(-0.1, 0.1, size=(n, dims))
        for fb, tol in ((8, 1e-10), (4, 2e-4)):
            eng, orc = engines(p, s, fb)
            d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
            ie, io = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
            np.testing.assert_allclose(d1[ie], d2[io], rtol=0, atol=tol * max(np.abs(d2).max(), 1e-300), err_msg=f"drho {dims}-D n={n} fb={fb}")
            np.testing.assert_allclose(a1[ie], a2[io], rtol=0, atol=tol * max(np.abs(a2).max(), 1e-300), err_msg=f"acc {dims}-D n={n} fb={fb}")
        eng, orc = engines(p, s, 8)
        pe, po = eng.advance(1e9, max_steps=3), orc.advance(1e9, max_steps=3)
        assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
        e, o = by_id(eng.download()), by_id(orc.download())
        assert relmax(e["Density"], o["Density"]) < 1e-9
        assert np.abs(e["Position"] - o["Position"]).max() <= 1e-10 * np.abs(o["Position"]).max()
