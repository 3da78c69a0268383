"""Float32 HOST arrays (`host_float_bytes = 4`): a `SimulationMetaData{D,Float32,…}` user of the reference.

The Julia shim sends `sizeof(T)` as host_float_bytes (julia/SPHExampleMI355X.jl: open_session), so a Float32 simulation lands on the
Float32 branches of upload (pack_host<float>), download (download_begin_as<float>, k_pack_output), forces_once, the kernel output and
the slab driver's split/merge.  Until round 6 every test passed Float64 arrays.  The reference itself computes such a run in mixed
precision (`clamp(…, 0.0, 2.0)` and the 0.5 of the viscosity promote to Float64, src/SPHCellList.jl:280,
src/SPHViscosityModels.jl:66), so the parity target is the fp64 oracle FED THE SAME ROUNDED INPUTS.

Tolerances, stated: inputs and outputs are fp32, so 1e-6 relative on density and position is the honest bar for the STATE (half an ulp
of fp32 is 6e-8; twenty steps of fp32 kernels with double-float state stay below 1e-6 on these cases, the fp64 kernels at the output
rounding — measured: fp64 kernels 6e-8, fp32 kernels 1.1e-6 on the 2-D dam break, so the fp32-kernel bar is 3e-6, inside the north
star's 1e-5); single force evaluations: the error class of tests/test_engine_gpu.py (fp32 cancellation in a sum of O(100) pair terms,
2e-4 of the field maximum there; 2.5e-4 measured on these rounded inputs: bar 4e-4) for fp32 kernels, and for fp64 kernels 2e-7 — the
result is rounded to the caller's Float32."""
import numpy as np
import pytest

from sphexample_amd.preprocess import FIELD_NAMES, SimParticles

pytestmark = pytest.mark.gpu


def as_float32(p) -> SimParticles:
    q = p.copy()
    q.FloatType = np.float32
    for k in FIELD_NAMES:
        a = getattr(q, k)
        if a.dtype == np.float64:
            setattr(q, k, np.ascontiguousarray(a.astype(np.float32)))
    if getattr(p, "geometries", None) is not None:
        q.geometries = p.geometries
    return q


def by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


def relmax(a, b):
    return np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300)


def engines(p32, s, fb, **kw):
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    eng = make_engine(p32, s, device_float_bytes=fb, **kw)
    assert eng.cfg.host_float_bytes == 4
    return eng, make_oracle(p32, s)          # the oracle's upload widens the SAME fp32 values to fp64


@pytest.mark.parametrize("case", ["dam_break_2d", "dam_break_3d_shipped", "still_wedge"])
@pytest.mark.parametrize("fb", [4, 8])
def test_upload_download_round_trip_is_exact(case, fb, request):
    """What goes up as Float32 comes down as the same Float32 bits (fp32 kernels: record = the value, low word 0; fp64 kernels: widened
    and rounded back), in Float32 arrays, for every field the boundary carries."""
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    p32 = as_float32(p)
    rng = np.random.default_rng(5)
    p32.Velocity = rng.standard_normal(p32.Velocity.shape).astype(np.float32)
    p32.Acceleration = rng.standard_normal(p32.Acceleration.shape).astype(np.float32)
    eng = make_engine(p32, s, device_float_bytes=fb)
    d = eng.download()
    for k in ("Position", "Velocity", "Acceleration", "Density", "Pressure", "GhostPoints"):
        assert d[k].dtype == np.float32, k
    got = by_id(d)
    o = np.argsort(p32.ID, kind="stable")
    for k in ("Position", "Velocity", "Acceleration", "Density", "GhostPoints"):
        np.testing.assert_array_equal(got[k], getattr(p32, k)[o], err_msg=k)
    np.testing.assert_array_equal(got["Type"], p32.Type[o])
    np.testing.assert_array_equal(got["GroupMarker"], p32.GroupMarker[o])


@pytest.mark.parametrize("case", ["dam_break_2d", "dam_break_3d_shipped"])
@pytest.mark.parametrize("fb,tol", [(8, 2e-7), (4, 4e-4)])
def test_single_force_evaluation_from_float32_arrays(case, fb, tol, request):
    from conftest import perturbed
    p, s = request.getfixturevalue(case)
    p32 = as_float32(perturbed(p, seed=11))
    eng, orc = engines(p32, s, fb)
    d1, a1 = eng.forces_once()
    d2, a2 = orc.forces_once()
    assert d1.dtype == np.float32 and a1.dtype == np.float32
    e, o = eng.download(), orc.download()
    ie, io = np.argsort(e["ID"], kind="stable"), np.argsort(o["ID"], kind="stable")
    if fb == 8:
        np.testing.assert_array_equal(e["ID"], o["ID"])
        np.testing.assert_array_equal(e["Cells"], o["Cells"])
    assert relmax(d1[ie], d2[io]) < tol
    assert relmax(a1[ie], a2[io]) < tol


@pytest.mark.parametrize("case,steps", [("dam_break_2d", 20), ("dam_break_3d_shipped", 20)])
@pytest.mark.parametrize("fb", [8, 4])
def test_twenty_steps_from_float32_arrays(case, steps, fb, request):
    p, s = request.getfixturevalue(case)
    p32 = as_float32(p)
    eng, orc = engines(p32, s, fb)
    pe, po = eng.advance(1e9, max_steps=steps), orc.advance(1e9, max_steps=steps)
    assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-9 if fb == 8 else 1e-5)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert e["Density"].dtype == np.float32 and e["Position"].dtype == np.float32
    np.testing.assert_array_equal(e["ID"], o["ID"])
    err_rho, err_x = relmax(e["Density"], o["Density"]), np.abs(e["Position"].astype(np.float64) - o["Position"]).max() / np.abs(o["Position"]).max()
    print(f"[host fp32] {case} fb={fb}: rho {err_rho:.2e} x {err_x:.2e}")
    bar = 1e-6 if fb == 8 else 3e-6
    assert err_rho < bar and err_x < bar, (err_rho, err_x)
    if fb == 8:
        assert pe.index_counter == po.index_counter
        vmax = max(np.abs(o["Velocity"]).max(), 1e-12)
        assert np.abs(e["Velocity"].astype(np.float64) - o["Velocity"]).max() / vmax < 1e-6


def test_mdbc_from_float32_arrays(still_wedge):
    """C5's layout with Float32 arrays: GhostPoints cross the boundary as Float32 too (fp64 kernels: the policy's choice for mDBC)."""
    p, s = still_wedge
    p32 = as_float32(p)
    eng, orc = engines(p32, s, 0)
    assert eng.device_float_bytes == 8
    pe, po = eng.advance(1e9, max_steps=30), orc.advance(1e9, max_steps=30)
    assert (pe.iteration, pe.n_rebuilds, pe.index_counter) == (po.iteration, po.n_rebuilds, po.index_counter)
    e, o = by_id(eng.download()), by_id(orc.download())
    assert relmax(e["Density"], o["Density"]) < 1e-6
    assert np.abs(e["Position"].astype(np.float64) - o["Position"]).max() < 1e-6 * np.abs(o["Position"]).max()


@pytest.mark.parametrize("fb", [4, 8])
def test_slabs_from_float32_arrays(dam_break_3d_shipped, fb):
    """The slab driver splits the caller's Float32 arrays (MultiEngine::upload takes host_float_bytes-wide rows) and merges Float32
    rows on the way out: three slabs against the one-device handle, same IDs in the same order, state within the precision's tolerance."""
    from sphexample_amd.engine import make_engine
    p, s = dam_break_3d_shipped
    p32 = as_float32(p)
    one = make_engine(p32, s, device_float_bytes=fb)
    dd = make_engine(p32, s, device_float_bytes=fb, devices=[0, 0, 0])
    p1, p3 = one.advance(1e9, max_steps=30), dd.advance(1e9, max_steps=30)
    assert (p1.iteration, p1.n_rebuilds, p1.index_counter) == (p3.iteration, p3.n_rebuilds, p3.index_counter)
    a, b = one.download(), dd.download()
    assert b["Position"].dtype == np.float32
    np.testing.assert_array_equal(a["ID"], b["ID"])
    np.testing.assert_array_equal(a["Cells"], b["Cells"])
    assert relmax(b["Density"], a["Density"].astype(np.float64)) < 1e-6
    assert relmax(b["Position"], a["Position"].astype(np.float64)) < 1e-6


def test_kernel_output_into_float32_arrays(dam_break_2d_variants):
    """StoreKernelOutput (src/SPHCellList.jl:106-116) downloaded into Float32 arrays equals the Float64 download rounded."""
    import dataclasses
    from sphexample_amd import StoreKernelOutput
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d_variants
    s = dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, KMode=StoreKernelOutput))
    e64 = make_engine(p, s, device_float_bytes=8)
    e32 = make_engine(as_float32(p), s, device_float_bytes=8)
    e64.advance(1e9, max_steps=1); e32.advance(1e9, max_steps=1)
    k64, g64 = e64.kernel_output()
    k32, g32 = e32.kernel_output()
    assert k32.dtype == np.float32 and g32.dtype == np.float32
    # (rows are in each handle's own cell-sorted order, and positions rounded to fp32 need not sort the same way: compare by ID)
    o64, o32 = np.argsort(e64.download(("ID",))["ID"], kind="stable"), np.argsort(e32.download(("ID",))["ID"], kind="stable")
    k64, g64, k32, g32 = k64[o64], g64[o64], k32[o32], g32[o32]
    # (the two handles start from inputs that differ by the fp32 rounding of the positions: 1e-5 of the kernel sum is that, not the download)
    assert relmax(k32, k64) < 1e-5 and relmax(g32, g64) < 1e-4
