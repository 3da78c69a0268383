"""The slab driver inside libsphmi.so (csrc/sphmi_multi.h) behind the ordinary sphmi_create / sphmi_advance.

CPU part: the host-side planner (axis, cuts by work, halo width, capacities) through sphmi_plan_slabs — no device needed.
GPU part: a handle created with a device list — here the list repeats GPU 0, so the slabs share the one GPU of the test
box and their messages are stream-ordered device copies; the step sequence, the migration, the ghost layers, the order
tags, the re-cut and the device-side step control are the ones an 8-GPU run uses — must reproduce the single-device
handle: same dt sequence, same rebuild cadence, density / position to summation order (the tiles differ).
"""
import ctypes as C

import numpy as np
import pytest

from sphexample_amd._abi import make_config
from slab_planner_reference import SlabPlan, cell_x_of, choose_axis, particle_work


def _plan(p, s, world, fb=8, axis=None):
    from sphexample_amd.engine import load_library
    lib = load_library(rebuild_if_stale=False)
    cfg = make_config(len(p), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion,
                      device_float_bytes=fb, host_float_bytes=8)
    if axis is not None:
        cfg.slab_axis = axis + 1
    ax, hw = C.c_int32(), C.c_int32()
    cuts = (C.c_int64 * 16)(); owned = (C.c_int64 * 16)(); cap = (C.c_int64 * 16)()
    pos = np.ascontiguousarray(p.Position, dtype=np.float64)
    gp = np.ascontiguousarray(p.GhostPoints, dtype=np.float64) if cfg.mdbc else None
    lib.sphmi_plan_slabs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5
    rc = lib.sphmi_plan_slabs(C.byref(cfg), pos.ctypes.data_as(C.c_void_p), None if gp is None else gp.ctypes.data_as(C.c_void_p),
                              len(p), world, C.byref(ax), C.byref(hw), cuts, owned, cap)
    assert rc == 0, rc
    return ax.value, hw.value, list(cuts)[:world - 1], list(owned)[:world], list(cap)[:world]


@pytest.mark.parametrize("world", [2, 4])
def test_planner_matches_the_python_reference_plan(dam_break_3d_shipped, world):
    """sphmi_plan_slabs = slab_planner_reference.py's choose_axis + SlabPlan.from_columns (work-balanced exact cuts)."""
    p, s = dam_break_3d_shipped
    cols = [cell_x_of(p.Position[:, a], s.SimKernel.H_inv) for a in range(3)]
    w = particle_work(cols)
    ax, hw, cuts, owned, cap = _plan(p, s, world)
    assert hw == 1
    assert ax == choose_axis(cols, world, 2, w)
    ref = SlabPlan.from_columns(cols[ax], world, 2, w)
    assert cuts == ref.cuts()
    assert owned == list(np.bincount(ref.owner_of(cols[ax]), minlength=world))
    assert sum(owned) == len(p) and all(c > o for c, o in zip(cap, owned))
    for a in range(3):
        ax2, _, cuts2, _, _ = _plan(p, s, 2, axis=a)
        assert ax2 == a and cuts2 == SlabPlan.from_columns(cols[a], 2, 2, w).cuts()


def test_planner_mdbc_halo_width(dam_break_2d_mdbc, still_wedge):
    """Ghost nodes sit up to `off` columns from their boundary particle: ghost layers 2 + off columns wide, slabs never
    narrower than that."""
    p, s = dam_break_2d_mdbc
    ax, hw, cuts, owned, cap = _plan(p, s, 2)
    assert hw == 5 and sum(owned) == len(p)
    ax, hw, cuts, owned, cap = _plan(*still_wedge, 2)
    assert hw >= 3


def test_planner_rejects_too_many_slabs(dam_break_2d):
    from sphexample_amd.engine import load_library
    p, s = dam_break_2d
    lib = load_library(rebuild_if_stale=False)
    cfg = make_config(len(p), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion)
    pos = np.ascontiguousarray(p.Position, dtype=np.float64)
    lib.sphmi_plan_slabs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5
    cfg.slab_axis = 2                                     # y: 39 cell columns host 16 slabs of ≥ 2 columns …
    assert lib.sphmi_plan_slabs(C.byref(cfg), pos.ctypes.data_as(C.c_void_p), None, len(p), 16, None, None, None, None, None) == 0
    cfg.k, cfg.H_inv = 4.0, cfg.H_inv / 2                 # … but not with cells twice as wide (20 columns)
    assert lib.sphmi_plan_slabs(C.byref(cfg), pos.ctypes.data_as(C.c_void_p), None, len(p), 16, None, None, None, None, None) != 0
    assert lib.sphmi_plan_slabs(C.byref(cfg), pos.ctypes.data_as(C.c_void_p), None, len(p), 17, None, None, None, None, None) != 0


# ---------------------------------------------------------------------------------------------------------------------
def _by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


def _compare(case, steps, fb, tol, axis, world, request, calls=1, cuts_shift=0, env=None, monkeypatch=None):
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    ref = make_engine(p, s, device_float_bytes=fb)
    cuts = None
    if cuts_shift:
        cx = cell_x_of(p.Position[:, axis], s.SimKernel.H_inv)
        cuts = [c + cuts_shift for c in SlabPlan.from_columns(cx, world).cuts()]
    dd = make_engine(p, s, device_float_bytes=fb, devices=[0] * world, slab_axis=axis, cuts=cuts)
    for _ in range(calls):
        pr = ref.advance(1e9, max_steps=steps // calls)
        pd = dd.advance(1e9, max_steps=steps // calls)
        assert (pd.iteration, pd.steps_done, pd.n_rebuilds) == (pr.iteration, pr.steps_done, pr.n_rebuilds)
        assert pd.total_time == pytest.approx(pr.total_time, rel=1e-12 if fb == 8 else 1e-6)
        assert pd.last_dt == pytest.approx(pr.last_dt, rel=1e-12 if fb == 8 else 1e-5)
        assert pd.index_counter == pr.index_counter
    info = dd.multi_info()
    assert info.world == world and info.n_local == world and info.transport == 0
    assert axis is None or info.axis == axis
    from sphexample_amd.config import SimpleMDBC
    assert (info.halo_width >= 3) if s.SimMetaData.BMode is SimpleMDBC else (info.halo_width == 1)
    r = ref.download(("Position", "Density", "ID", "Velocity", "Cells"))
    d = dd.download(("Position", "Density", "ID", "Velocity", "Cells"))
    assert dd.owned_count() == len(p)
    # the merged download is in the order ONE engine holds: cell-sorted, in-cell order by history (order tags)
    np.testing.assert_array_equal(d["ID"], r["ID"])
    np.testing.assert_array_equal(d["Cells"], r["Cells"])
    assert np.abs(d["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < tol
    assert np.abs(d["Position"] - r["Position"]).max() / np.abs(r["Position"]).max() < tol
    np.testing.assert_array_equal(dd.unique_cells(), ref.unique_cells())
    return info


@pytest.mark.gpu
@pytest.mark.parametrize("case,steps,fb,tol,axis", [
    ("dam_break_3d_shipped", 30, 8, 1e-9, None), ("dam_break_3d_shipped", 30, 4, 1e-5, None),
    ("dam_break_3d_shipped", 30, 8, 1e-9, 0), ("dam_break_3d_shipped", 30, 8, 1e-9, 1), ("dam_break_3d_shipped", 30, 8, 1e-9, 2),
    ("dam_break_2d", 60, 8, 1e-9, None), ("dam_break_2d", 60, 8, 1e-9, 0), ("dam_break_2d", 60, 8, 1e-9, 1),
    ("dam_break_2d_variants", 40, 8, 1e-9, None),
    # a Moving body crossing nothing / the cut; 150 steps along y: fluid pushed by the body crosses the cut particle by
    # particle, and a migrant must take the in-cell place its previous GLOBAL sorted index gives it (order tags)
    ("moving_square", 40, 8, 1e-9, 0), ("moving_square", 150, 8, 1e-9, 1),
    # mDBC: ghost layers 2 + off columns wide, ghost copies corrected locally
    ("dam_break_2d_mdbc", 40, 8, 1e-9, None), ("dam_break_2d_mdbc", 40, 8, 1e-9, 0), ("dam_break_2d_mdbc", 40, 8, 1e-9, 1),
    ("dam_break_2d_mdbc", 40, 4, 2e-5, None), ("still_wedge", 40, 8, 1e-9, None), ("duckling", 12, 8, 1e-9, None)])
def test_two_slabs_in_one_handle_match_one_device(case, steps, fb, tol, axis, request):
    _compare(case, steps, fb, tol, axis, 2, request)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [3, 4, 8])
def test_more_slabs_in_one_handle(world, request):
    """Middle slabs have two neighbours (two ghost layers, two messages per pass, migration both ways)."""
    _compare("dam_break_3d_shipped", 40, 8, 1e-9, None, world, request)


@pytest.mark.gpu
@pytest.mark.parametrize("poison", ["255", "127"])
@pytest.mark.parametrize("world", [2, 3])
def test_rows_never_written_are_never_read(world, poison, monkeypatch):
    """The interior launch of a slab runs BEFORE the halo of its pass has landed.  A tile of a sparse cloud may hold ghost rows
    next to owned ones (an empty boundary cell: the sort puts the ghost column of one cell row next to the first cells of the
    next), and what a ghost row holds at that moment is last step's state — or, in a record set no pass has written yet,
    whatever the allocation held.  A NaN there used to become the tile's origin and reach, and every lane of the tile lost
    (or gained) all its candidates by the sign of that NaN: intermittent wrong densities in the first steps of a new handle,
    found by a fresh generation of tests/test_fuzz_gpu.py.  $SPHMI_POISON fills the record sets with NaNs of either sign at
    creation, so the read shows in every run: a random 2-D cloud on 2 and 3 slabs against the one-device handle, one call of
    five steps (a batch: the sets rotate without a rebuild in between)."""
    from sphexample_amd import (LinearDensityDiffusion, SimulationConstants, SimulationMetaData, SPHKernelInstance, WendlandC2,
                                ZeroViscosity, particles_from_arrays)
    from sphexample_amd.cases import CaseSetup
    from sphexample_amd.engine import make_engine
    rng = np.random.default_rng(8005)
    n, dx = 6000, 0.02
    side = n ** 0.5 * dx
    pos = rng.uniform(0, side, size=(n, 2)) - 50.0
    typ = rng.choice([1, 1, 1, 2, 3], size=n).astype(np.uint8)
    p = particles_from_arrays(2, pos, 1000.0 + rng.uniform(-5, 15, n), typ, rng.integers(1, 4, n), rng.permutation(n) + 1)
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, 2))
    s = CaseSetup("sparse cloud", SimulationConstants(dx=dx, m0=1000 * dx ** 2, c0=40.0, alpha=0.01, g=9.81, CFL=0.2),
                  SPHKernelInstance(2, WendlandC2(), dx=dx, k=2.0), SimulationMetaData(Dimensions=2), ZeroViscosity(), LinearDensityDiffusion())
    ref = make_engine(p, s, device_float_bytes=8)
    pr = ref.advance(1e9, max_steps=5)
    r = _by_id(ref.download())
    monkeypatch.setenv("SPHMI_POISON", poison)
    dd = make_engine(p, s, device_float_bytes=8, devices=[0] * world)
    pd = dd.advance(1e9, max_steps=5)
    assert (pd.iteration, pd.n_rebuilds, pd.index_counter) == (pr.iteration, pr.n_rebuilds, pr.index_counter)
    d = _by_id(dd.download())
    assert np.isfinite(d["Density"]).all()
    assert np.abs(d["Density"] - r["Density"]).max() < 1e-9 * np.abs(r["Density"]).max()
    assert np.abs(d["Position"] - r["Position"]).max() < 1e-12 * np.abs(r["Position"]).max()


@pytest.mark.gpu
def test_serial_halo_path(request, monkeypatch):
    """SPHMI_DD_OVERLAP=0: halo → whole pass on one stream (the path mDBC's pass 1 always takes)."""
    _compare("dam_break_3d_shipped", 30, 8, 1e-9, 1, 2, request, env={"SPHMI_DD_OVERLAP": "0"}, monkeypatch=monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("case,world", [("dam_break_3d_shipped", 3), ("moving_square", 2)])
def test_one_slab_at_a_time_measurement_mode_changes_no_result(case, world, request, monkeypatch):
    """SPHMI_DD_ONE_SLAB_AT_A_TIME=1 (tools/slab_pass_time.py: every slab's pass alone on the chip, timed by the host — the input of DESIGN §7's
    scaling prediction): a different ORDER of the same launches and copies, so the state must be the default path's, and the hook reports times."""
    info = _compare(case, 30, 8, 1e-9, None, world, request, env={"SPHMI_DD_ONE_SLAB_AT_A_TIME": "1"}, monkeypatch=monkeypatch)
    assert info.world == world


@pytest.mark.gpu
def test_a_second_upload_on_a_multi_slab_handle(dam_break_3d_shipped):
    """sphmi_upload makes new slab engines; the collective rebuild that follows must not ask them for the work histogram of a cell list they do not
    have yet (round 6: it did — "sphmi_dd_column_cost before the first rebuild" — whenever the handle had rebuilt before).  Same state as a one-device
    handle that is uploaded twice."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_3d_shipped
    one, dd = make_engine(p, s, device_float_bytes=8), make_engine(p, s, device_float_bytes=8, devices=[0, 0, 0])
    for e in (one, dd):
        e.advance(1e9, max_steps=7)
        e.upload_particles(p)
    pr, pd = one.advance(1e9, max_steps=30), dd.advance(1e9, max_steps=30)
    assert (pd.steps_done, pd.index_counter) == (pr.steps_done, pr.index_counter) and pd.last_dt == pytest.approx(pr.last_dt, rel=1e-12)
    a, b = one.download(("ID", "Density", "Position")), dd.download(("ID", "Density", "Position"))
    np.testing.assert_array_equal(a["ID"], b["ID"])
    assert np.abs(a["Density"] - b["Density"]).max() < 1e-9 * np.abs(a["Density"]).max()
    assert np.abs(a["Position"] - b["Position"]).max() < 1e-9 * np.abs(a["Position"]).max()


@pytest.mark.gpu
def test_cuts_move_with_the_work(request):
    """Start four columns off balance: the first rebuilds move the cuts back (particles migrate, ghost layers and halo
    lists are rebuilt) and the result is still the one-device one."""
    info = _compare("dam_break_3d_shipped", 80, 8, 1e-9, 0, 2, request, calls=2, cuts_shift=4)
    assert info.n_recuts >= 1


@pytest.mark.gpu
def test_cuts_move_with_the_work_mdbc(request, monkeypatch):
    info = _compare("dam_break_2d_mdbc", 80, 8, 1e-9, 0, 2, request, calls=2, cuts_shift=3, env={"SPHMI_DD_RECUT": "1.02"}, monkeypatch=monkeypatch)
    assert info.n_recuts >= 1 and info.halo_width == 5


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["dam_break_2d", "dam_break_3d_shipped"])
def test_output_intervals_keep_the_reductions(case, request):
    """Several SimulationLoop calls (advance to a TIME, as the reference's driver does): the control that ends an
    interval returns before it has used the maxima of the last corrector — they must still be there for the first Δt
    of the next interval, or the multi-device run takes a different first step than the one-device handle."""
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    ref = make_engine(p, s, device_float_bytes=8)
    dd = make_engine(p, s, device_float_bytes=8, devices=[0, 0])
    dt0 = 0.2 * s.SimKernel.h / s.SimConstants.c0
    for k in range(1, 5):
        pr, pd = ref.advance(7.3 * k * dt0), dd.advance(7.3 * k * dt0)
        assert (pd.iteration, pd.steps_done, pd.n_rebuilds) == (pr.iteration, pr.steps_done, pr.n_rebuilds)
        assert pd.last_dt == pytest.approx(pr.last_dt, rel=1e-12)
        assert pd.total_time == pytest.approx(pr.total_time, rel=1e-12)
    r, d = _by_id(ref.download(("ID", "Density"))), _by_id(dd.download(("ID", "Density")))
    assert np.abs(d["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < 1e-9


@pytest.mark.gpu
def test_rccl_binds_and_a_one_rank_world_runs(dam_break_3d_shipped):
    """sphmi_create_rank with world = 1: RCCL is bound (librccl.so.1 resolved, unique id made), the slab driver runs its
    full step sequence with no peers, and the slab's download is the whole particle set.  (More than one RCCL rank
    needs more than one GPU: the 8-GPU run is the driver's.)"""
    from sphexample_amd.engine import make_engine, rccl_unique_id
    p, s = dam_break_3d_shipped
    uid = rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    ref = make_engine(p, s, device_float_bytes=8)
    dd = make_engine(p, s, device_float_bytes=8, rank=0, world=1, unique_id=uid)
    pr, pd = ref.advance(1e9, max_steps=20), dd.advance(1e9, max_steps=20)
    assert (pd.iteration, pd.n_rebuilds, pd.total_time) == (pr.iteration, pr.n_rebuilds, pr.total_time)
    r, d = ref.download(("ID", "Density")), dd.download(("ID", "Density"))
    np.testing.assert_array_equal(d["ID"], r["ID"])
    assert np.abs(d["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < 1e-12


# ---- the hooks single-device handles have, on multi-device handles --------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case,world,mdbc", [("dam_break_3d_shipped", 2, False), ("dam_break_3d_shipped", 3, False),
                                              ("dam_break_2d_mdbc", 2, True), ("duckling", 2, True)])
def test_forces_once_on_slabs(case, world, mdbc, request):
    """sphmi_forces_once of a multi-device handle: collective rebuild (fresh ghost layers), Pressure!, [mDBC], one forces-only
    pass over interior + edge tiles, rows merged into the one-device order.  Then the run goes on as if nothing happened."""
    from conftest import perturbed
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    q = perturbed(p, seed=9)
    ref, dd = make_engine(q, s, device_float_bytes=8), make_engine(q, s, device_float_bytes=8, devices=[0] * world)
    d1, a1 = ref.forces_once(apply_mdbc=mdbc)
    d2, a2 = dd.forces_once(apply_mdbc=mdbc)
    np.testing.assert_array_equal(dd.download(("ID",))["ID"], ref.download(("ID",))["ID"])
    assert np.abs(d2 - d1).max() <= 1e-10 * np.abs(d1).max() and np.abs(a2 - a1).max() <= 1e-10 * np.abs(a1).max()
    pr, pd = ref.advance(1e9, max_steps=12), dd.advance(1e9, max_steps=12)
    assert (pd.iteration, pd.n_rebuilds, pd.index_counter) == (pr.iteration, pr.n_rebuilds, pr.index_counter)
    r, d = ref.download(("ID", "Density")), dd.download(("ID", "Density"))
    np.testing.assert_array_equal(d["ID"], r["ID"])
    assert np.abs(d["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_kernel_output_on_slabs(dam_break_2d, world):
    """StoreKernelOutput (src/SPHCellList.jl:106-116) on a multi-device handle: ΣW, Σ∇W of the last corrector, merged."""
    import dataclasses
    from conftest import perturbed
    from sphexample_amd import StoreKernelOutput
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    s = dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, KMode=StoreKernelOutput))
    q = perturbed(p, seed=3)
    ref, dd = make_engine(q, s, device_float_bytes=8), make_engine(q, s, device_float_bytes=8, devices=[0] * world)
    ref.advance(1e9, max_steps=20); dd.advance(1e9, max_steps=20)
    (k1, g1), (k2, g2) = ref.kernel_output(), dd.kernel_output()
    np.testing.assert_array_equal(dd.download(("ID",))["ID"], ref.download(("ID",))["ID"])
    assert k1.max() > 0 and np.abs(k2 - k1).max() <= 1e-10 * np.abs(k1).max() and np.abs(g2 - g1).max() <= 1e-9 * np.abs(g1).max()


@pytest.mark.gpu
def test_async_download_on_slabs(dam_break_3d_shipped):
    """sphmi_download_begin / _end of a multi-device handle: the snapshot is taken in stream order on every slab, the copies
    run while the next interval is computed, `end` merges — the arrays hold the state AT THE BEGIN call."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_3d_shipped
    ref, dd = make_engine(p, s, device_float_bytes=8), make_engine(p, s, device_float_bytes=8, devices=[0, 0, 0])
    ref.advance(1e9, max_steps=25); dd.advance(1e9, max_steps=25)
    want = ref.download()
    snap = p.copy()
    dd.pin(snap)                                            # accepted (a no-op for multi-device handles)
    dd.download_into_begin(snap)
    dd.advance(1e9, max_steps=25)                           # incl. a collective rebuild: the snapshot must not move
    dd.download_end()
    dd.unpin()
    for k in ("ID", "Position", "Velocity", "Density", "Pressure", "Type", "GroupMarker", "Cells"):
        got = getattr(snap, k)
        if want[k].dtype.kind == "f":
            assert np.abs(got - want[k]).max() <= 1e-9 * max(np.abs(want[k]).max(), 1e-300), k
        else:
            np.testing.assert_array_equal(got, want[k], err_msg=k)
    ref.advance(1e9, max_steps=25)
    r, d = ref.download(("ID", "Density")), dd.download(("ID", "Density"))
    np.testing.assert_array_equal(d["ID"], r["ID"])
    assert np.abs(d["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("case,axis", [("dam_break_3d_shipped", 1), ("dam_break_2d", 0)])
def test_column_work_on_device_matches_host(case, axis, request):
    """The re-cut balances what the first cut balanced: the slabs' device-side work histogram (owned particles of every
    slab, cell list of the rebuild; test hook sphmi_multi_column_cost) = particle_work summed per column, exactly."""
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    cols = [cell_x_of(p.Position[:, a], s.SimKernel.H_inv) for a in range(p.Position.shape[1])]
    w = particle_work(cols)
    dd = make_engine(p, s, device_float_bytes=8, devices=[0, 0], slab_axis=axis)
    dd.advance(1e9, max_steps=1)                            # the first rebuild: cell list, ghost layers
    col0, ncols = int(cols[axis].min()), int(cols[axis].max() - cols[axis].min() + 1)
    got = np.zeros(ncols, dtype=np.uint64)
    dd._lib.sphmi_multi_column_cost.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    dd._check(dd._lib.sphmi_multi_column_cost(dd._h, col0, ncols, got.ctypes.data_as(C.c_void_p)))
    want = np.bincount(cols[axis] - col0, weights=w, minlength=ncols).astype(np.uint64)
    np.testing.assert_array_equal(got, want)
    assert want.sum() > 0
