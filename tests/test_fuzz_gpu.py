"""Seeded fuzzing of the engine against the oracle (fp64 kernels, so that every difference beyond summation order is a bug).

The shipped layouts are lattices in a box near the origin.  These cases are not: random clouds in 2-D and 3-D drawn as a
box, a thin sheet, a line, two far-apart clusters (cell grids with huge empty stretches: chunk skipping, sparse tiles) or
a dense blob (hundreds of particles per cell: the per-lane queues run full inside a chunk row), anywhere in space
(offsets of ±50 m: cells far from 0, negative cells), with random particle types, cut-offs k ∈ {1.5, 2, 2.5} and random
model tags — on one device and, when there are enough particles, on slabs.  Checked: same sorted order ID for ID, one force
evaluation to 1e-10 of the field maximum, the loop counters and the state after a few steps.
"""
import dataclasses
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

SHAPES = ("box", "sheet", "line", "clusters", "blob")
# $SPHMI_FUZZ_SEED0 shifts every seed: `SPHMI_FUZZ_SEED0=1000 pytest tests/test_fuzz_gpu.py` is a fresh generation of the same tests
SEED0 = int(os.environ.get("SPHMI_FUZZ_SEED0", "0"))


def _case(seed):
    from sphexample_amd import (ArtificialViscosity, Laminar, LaminarSPS, LinearDensityDiffusion, SimulationConstants,
                                SimulationMetaData, SPHKernelInstance, WendlandC2, ZeroViscosity, particles_from_arrays)
    from sphexample_amd.cases import CaseSetup
    from sphexample_amd.config import ComplexDensityDiffusion, ZeroGravityLinearDensityDiffusion
    rng = np.random.default_rng(1000 + seed)
    dims = int(rng.choice([2, 3]))
    shape = SHAPES[seed % len(SHAPES)]
    n = int(rng.choice([1, 7, 65, 200, 777, 777, 2500, 2500, 6000, 6000]))
    dx = float(rng.choice([0.01, 0.02, 0.05]))
    k = float(rng.choice([1.5, 2.0, 2.0, 2.5]))
    side = max(n, 8) ** (1.0 / dims) * dx
    if shape == "box":
        pos = rng.uniform(0, side, size=(n, dims))
    elif shape == "sheet":
        pos = rng.uniform(0, side * 2, size=(n, dims)); pos[:, -1] = rng.uniform(0, 1.5 * dx, size=n)
    elif shape == "line":
        pos = rng.uniform(0, 0.8 * dx, size=(n, dims)); pos[:, 0] = rng.uniform(0, n * dx * 0.6 + dx, size=n)
    elif shape == "clusters":
        pos = rng.uniform(0, side * 0.7, size=(n, dims)); pos[n // 2:, 0] += 40 * side + 3.0
    else:
        pos = rng.normal(0, 1.2 * dx, size=(n, dims))
    pos += rng.choice([0.0, 50.0, -13.7, 1e-9]) * rng.choice([1.0, -1.0], size=dims)
    typ = rng.choice([1, 1, 1, 2, 3], size=n).astype(np.uint8)
    p = particles_from_arrays(dims, pos, 1000.0 + rng.uniform(-5, 15, n), typ, rng.integers(1, 4, n), rng.permutation(n) + 1)
    p.Velocity[:] = rng.uniform(-1, 1, size=(n, dims))
    sc = SimulationConstants(dx=dx, m0=1000 * dx ** dims, c0=float(rng.choice([20.0, 40.0, 90.0])), alpha=float(rng.choice([1e-6, 0.01, 0.1])),
                             g=float(rng.choice([9.81, 0.0])), CFL=0.2)
    ker = SPHKernelInstance(dims, WendlandC2(), dx=dx, k=k)
    visc = [ZeroViscosity(), ArtificialViscosity(), ArtificialViscosity(), Laminar(), LaminarSPS()][int(rng.integers(0, 5))]
    ddt = [LinearDensityDiffusion(), LinearDensityDiffusion(), ZeroGravityLinearDensityDiffusion(), ComplexDensityDiffusion()][int(rng.integers(0, 4))]
    s = CaseSetup(f"fuzz{seed}", sc, ker, SimulationMetaData(Dimensions=dims), visc, ddt)
    return p, s, shape


def _by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 40))
def test_fuzzed_cloud_matches_the_oracle(seed):
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    p, s, shape = _case(seed)
    n = len(p)
    eng, orc = make_engine(p, s, device_float_bytes=8), make_oracle(p, s)
    d1, a1 = eng.forces_once(); d2, a2 = orc.forces_once()
    np.testing.assert_array_equal(eng.download(("ID",))["ID"], orc.download(("ID",))["ID"])
    np.testing.assert_array_equal(eng.download(("Cells",))["Cells"], orc.download(("Cells",))["Cells"])
    np.testing.assert_array_equal(eng.unique_cells(), orc.unique_cells())
    np.testing.assert_allclose(d1, d2, rtol=0, atol=1e-10 * max(np.abs(d2).max(), 1e-300), err_msg=f"drho {shape}")
    np.testing.assert_allclose(a1, a2, rtol=0, atol=1e-10 * max(np.abs(a2).max(), 1e-300), err_msg=f"acc {shape}")
    eng, orc = make_engine(p, s, device_float_bytes=8), make_oracle(p, s)
    steps = 5
    po = orc.advance(1e9, max_steps=steps)
    # A violent cloud (the dense blobs) may drive a density through zero.  The reference has no check and carries on with a
    # negative density; the engine keeps the MotionLimiter flag in the sign of ρ, so it must REFUSE (SPHMI_ERR_NUMERIC) — in the
    # very call that produced it — and must never hand such a state out as if it were good.
    bad = bool((orc.download(("Density",))["Density"] <= 0).any()) or not np.isfinite(po.last_dt)
    from sphexample_amd._abi import ERR_NUMERIC, SphmiError
    if bad:
        with pytest.raises(SphmiError) as ei:
            eng.advance(1e9, max_steps=steps)
        assert ei.value.status == ERR_NUMERIC
        return
    try:
        pe = eng.advance(1e9, max_steps=steps)
    except SphmiError as exc:
        # The oracle's FINAL state is sane, the engine refused: legitimate only if a density went through zero at an INTERMEDIATE step
        # and came back (the reference never looks).  The oracle stepped one step at a time shows it (the rebuild every call opens
        # with changes the summation order, not the physics).
        assert exc.status == ERR_NUMERIC, exc
        o1 = make_oracle(p, s)
        seen = False
        for _ in range(steps):
            o1.advance(1e9, max_steps=1)
            seen = seen or bool((o1.download(("Density",))["Density"] <= 0).any())
        assert seen, f"the engine refused a run in which no density of the oracle ever reaches zero ({shape}): {exc}"
        return
    assert (pe.iteration, pe.n_rebuilds, pe.index_counter) == (po.iteration, po.n_rebuilds, po.index_counter)
    assert pe.total_time == pytest.approx(po.total_time, rel=1e-9)
    e, o = _by_id(eng.download()), _by_id(orc.download())
    scale = max(np.abs(o["Position"] - o["Position"].mean(0)).max(), s.SimKernel.h)
    assert np.abs(e["Density"] - o["Density"]).max() < 1e-8 * np.abs(o["Density"]).max(), shape
    assert np.abs(e["Position"] - o["Position"]).max() < 1e-9 * scale + 1e-15 * np.abs(o["Position"]).max(), shape
    # Three more calls on the same handle (round 4): every call opens with a rebuild, and from the second rebuild of a handle on a
    # cloud of this size rebuilds on the device, on the grid of its first one + two cell layers — a cloud that flies apart leaves
    # that grid and sends the rebuild back to the host (Engine::rebuild_device, error 3).  Same counters and state as the oracle.
    ids5 = eng.download(("ID",))["ID"]
    if n >= 7:
        try:
            for _ in range(3):
                po2 = orc.advance(1e9, max_steps=3)
                if bool((orc.download(("Density",))["Density"] <= 0).any()) or not np.isfinite(po2.last_dt):
                    raise StopIteration
                pe2 = eng.advance(1e9, max_steps=3)
                assert (pe2.iteration, pe2.n_rebuilds, pe2.index_counter) == (po2.iteration, po2.n_rebuilds, po2.index_counter), shape
            e2, o2 = _by_id(eng.download()), _by_id(orc.download())
            np.testing.assert_array_equal(eng.download(("ID",))["ID"], orc.download(("ID",))["ID"])
            assert np.abs(e2["Density"] - o2["Density"]).max() < 1e-8 * np.abs(o2["Density"]).max(), shape
            assert np.abs(e2["Position"] - o2["Position"]).max() < 1e-9 * scale + 1e-15 * np.abs(o2["Position"]).max(), shape
        except StopIteration:
            pass                                        # a density went through zero on the way: covered by the refusal checks above
        except SphmiError as exc:
            assert exc.status == ERR_NUMERIC, exc
    if n >= 200 and shape in ("box", "sheet", "line", "clusters"):
        # the same on three slabs of one handle
        try:
            dd = make_engine(p, s, device_float_bytes=8, devices=[0, 0, 0])
        except Exception as exc:                        # too few cell columns for three slabs: a planning error, with a text
            assert "slab" in str(exc) or "devices" in str(exc) or "columns" in str(exc), exc
            return
        pd = dd.advance(1e9, max_steps=steps)
        assert (pd.iteration, pd.n_rebuilds, pd.index_counter) == (po.iteration, po.n_rebuilds, po.index_counter)
        d = dd.download()
        np.testing.assert_array_equal(d["ID"], ids5)
        dd_ = _by_id(d)
        assert np.abs(dd_["Density"] - o["Density"]).max() < 1e-8 * np.abs(o["Density"]).max(), shape


# ---- second generation: every switch behind the ABI at once ------------------------------------------------------------
def _case2(seed):
    """As _case, plus: CubicSpline kernel, PlanarShifting, StoreKernelOutput, mDBC with random ghost nodes, a Moving group with a
    MotionDetails, k = √2, particles sitting EXACTLY on cell faces, and fp32 or fp64 kernels."""
    from sphexample_amd import CubicSpline, Geometry, Moving, SPHKernelInstance, StoreKernelOutput, WendlandC2
    from sphexample_amd.config import MotionDetails, PlanarShifting, SimpleMDBC
    p, s, shape = _case(seed)
    rng = np.random.default_rng(5000 + seed)
    dims, n = s.SimMetaData.Dimensions, len(p)
    meta = s.SimMetaData
    kern = s.SimKernel
    if rng.random() < 0.3:
        kern = SPHKernelInstance(dims, CubicSpline(0.2), h=kern.h, k=kern.k)
    elif rng.random() < 0.3:
        kern = SPHKernelInstance(dims, WendlandC2(), h=kern.h, k=float(np.sqrt(2)))
    if rng.random() < 0.3:
        meta = dataclasses.replace(meta, SMode=PlanarShifting)
    if rng.random() < 0.3:
        meta = dataclasses.replace(meta, KMode=StoreKernelOutput)
    if rng.random() < 0.35 and shape != "blob":
        meta = dataclasses.replace(meta, BMode=SimpleMDBC)
        bnd = p.Type != 1
        p.GhostPoints[bnd] = p.Position[bnd] + rng.normal(0, 1.0, size=(int(bnd.sum()), dims)) * kern.h
    if rng.random() < 0.5:                                   # some particles exactly on cell faces (map_floor's tie rule)
        pick = rng.random(n) < 0.2
        H = kern.H
        p.Position[pick, 0] = np.round(p.Position[pick, 0] / H) * H + 0.5 * H * rng.choice([1.0, -1.0])
    if rng.random() < 0.4 and (p.Type == 3).any():
        p.GroupMarker[p.Type == 3] = 3
        d = np.zeros(dims); d[int(rng.integers(0, dims))] = 1.0
        p.geometries = [Geometry(CSVFile="", GroupMarker=3, Type=Moving, Motion=MotionDetails(Velocity=float(rng.uniform(0.5, 3)), StartTime=0.0,
                                                                                              Duration=float(rng.choice([1e-4, 10.0])), Direction=tuple(d)))]
    s = dataclasses.replace(s, SimKernel=kern, SimMetaData=meta)
    return p, s, shape, int(rng.choice([8, 8, 4]))


@pytest.mark.parametrize("seed", range(SEED0 + 100, SEED0 + 160))
def test_fuzzed_switches_match_the_oracle(seed):
    from oracle.oracle import make_oracle
    from sphexample_amd._abi import ERR_DOMAIN, ERR_NUMERIC, SphmiError
    from sphexample_amd.config import SimpleMDBC, StoreKernelOutput
    from sphexample_amd.engine import make_engine
    p, s, shape, fb = _case2(seed)
    mdbc = s.SimMetaData.BMode is SimpleMDBC
    tol_f = 1e-10 if fb == 8 else 1e-3
    if fb == 4:
        # fp32 handles keep ABSOLUTE coordinates in fp32: a cloud 50 m from the origin resolves 4·10⁻⁶ m, two random particles may sit
        # 10⁻⁴ m apart, and with k < 2 the kernel jumps at r = H — what an fp64 oracle fed with the ORIGINAL coordinates then shows is
        # the conditioning of the case, not the arithmetic of the kernels (seeds 13106, 14145, 16106: a handful of particles off by
        # a whole pair term).  So the case itself is rounded to fp32 first: engine and oracle start from the same numbers, the
        # oracle evaluates them exactly, and what is left is the engine's own arithmetic — tile-relative differences, rounded once
        # (6·10⁻⁸ of a few H) against the closest pair.
        for f in ("Position", "Velocity", "Density", "GhostPoints"):
            getattr(p, f)[...] = getattr(p, f).astype(np.float32).astype(np.float64)
        from scipy.spatial import cKDTree
        if len(p) > 1:
            dmin = cKDTree(p.Position).query(p.Position, k=2)[0][:, 1].min()
            tol_f *= 1.0 + 2000.0 * (8.0 * s.SimKernel.H * 6e-8) / max(dmin, 1e-300)
        # What remains after that (ten generations, the worst particles listed one by one): up to 2.5·10⁻³ of the field maximum in dρ/dt on a few
        # particles of the planes x = (n ± ½)·H the generator puts a fifth of the cloud on — hundreds of particles with EQUAL x,
        # pairs at 10⁻³ H, large terms of both signs — with or without the moving group, every particle type alike; the fp64
        # kernels hold 10⁻¹⁰ on the same cases, so the sums are the right sums.  fp32 on such a cloud is good to 4·10⁻³.
        tol_f *= 4.0
    what = f"{shape} fb{fb} {type(s.SimViscosity).__name__} {type(s.SimDensityDiffusion).__name__} {type(s.SimKernel.kernel).__name__} k{s.SimKernel.k:.2f} " \
           f"{s.SimMetaData.SMode.__name__} {s.SimMetaData.KMode.__name__} {s.SimMetaData.BMode.__name__} motion={getattr(p, 'geometries', None) is not None}"

    def both():
        e, o = make_engine(p, s, device_float_bytes=fb), make_oracle(p, s)
        if getattr(p, "geometries", None) is not None:
            o.set_motions(p.geometries)
        return e, o
    eng, orc = both()
    d1, a1 = eng.forces_once(apply_mdbc=mdbc); d2, a2 = orc.forces_once(apply_mdbc=mdbc)
    ie, io = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
    if fb == 8:
        np.testing.assert_array_equal(eng.download(("ID",))["ID"], orc.download(("ID",))["ID"], err_msg=what)
    def close(x, y, name):
        scale = max(np.abs(y).max(), 1e-300)
        err = np.abs(x - y).reshape(len(x), -1).max(axis=1) / scale
        if fb == 4 and (s.SimKernel.k < 2.0 or mdbc):
            # A kernel cut off BEFORE it vanishes (k < 2) switches a finite pair term at r = H (src/SPHCellList.jl:275), and fp32 rounds r² against H²: a pair
            # that sits on the cut within 1e-7 is in for one precision and out for the other — both of its particles then differ by ONE pair's term (found by
            # generation 117000, identical in round 4's tree: two particles of 2 500, 7e-3 of the field maximum).  That is why the library chooses fp64 kernels
            # for such handles (sphmi_auto_device_float_bytes); forced to fp32 here, a handful of particles — pairs — may sit outside, each by at most one pair's share.
            # The same for mDBC handles forced to fp32: ApplyMDBCCorrection (:598-622) takes one of its fallback branches for a lone neighbour at r ~ H or a
            # determinant next to 1e-3, a boundary particle then carries another density and its few neighbours another pair term (generation 182500, one
            # particle of 777 at 1.15 x the tolerance, identical in round 4's tree and with every waves-per-tile choice; tests/test_example_precision_gpu.py
            # holds the record behind the policy).
            bad = err > tol_f
            assert bad.sum() <= max(4, len(err) // 500) and err.max() < 0.05, f"{name} {what}: {int(bad.sum())} particles beyond {tol_f:.1e}, worst {err.max():.2e}"
            # … and every one of them must HAVE such a reason (round-5 advice: a flat allowance is no check) — a partner within 2e-5 of r = H (the pair that
            # is in for one precision and out for the other), or, on mDBC handles, a boundary particle (whose density ApplyMDBCCorrection may have taken from
            # the other branch) as itself or among its neighbours; positions of the state the forces were evaluated on, by ID
            if bad.any():
                from scipy.spatial import cKDTree
                st = orc.download(("ID", "Position", "Type"))
                oo = np.argsort(st["ID"], kind="stable")
                X, ty = st["Position"][oo], st["Type"][oo]
                H = s.SimKernel.H
                tree = cKDTree(X)
                for i in np.flatnonzero(bad):
                    nb = np.array(tree.query_ball_point(X[i], H * (1 + 2e-5)), dtype=np.int64)
                    nb = nb[nb != i]
                    r = np.linalg.norm(X[nb] - X[i], axis=1) if len(nb) else np.zeros(0)
                    on_cut = bool((np.abs(r - H) < 2e-5 * H).any())
                    near_boundary = mdbc and (ty[i] != 1 or bool((ty[nb] != 1).any()))
                    assert (s.SimKernel.k < 2.0 and on_cut) or near_boundary, \
                        f"{name} {what}: particle {i} is {err[i]:.2e} off and has neither a partner on r = H nor an mDBC boundary particle within reach"
            return
        np.testing.assert_allclose(x, y, rtol=0, atol=tol_f * scale, err_msg=name + " " + what)
    close(d1[ie], d2[io], "drho")
    close(a1[ie], a2[io], "acc")
    if fb == 4:
        return                                            # (K steps of a violent cloud in fp32: chaos, not parity)
    eng, orc = both()
    done = 0
    for steps in (2, 3):                                  # two calls: the rebuild that opens the second, the carried reductions
        po = orc.advance(1e9, max_steps=steps)
        bad = bool((orc.download(("Density",))["Density"] <= 0).any()) or not np.isfinite(po.last_dt) or not np.isfinite(orc.download(("Position",))["Position"]).all()
        if bad:
            with pytest.raises(SphmiError) as ei:
                eng.advance(1e9, max_steps=steps)
            assert ei.value.status == ERR_NUMERIC, what
            return
        try:
            pe = eng.advance(1e9, max_steps=steps)
        except SphmiError as exc:
            # The oracle's state after the call is sane and the engine refused: legitimate only if mDBC extrapolated a NON-POSITIVE
            # density to some boundary particle at the start of one of the steps (random ghost nodes do that; the reference carries
            # on with it and the boundary clamp at the end of the step hides it — the engine keeps the MotionLimiter flag in the
            # sign of ρ and must refuse).  The oracle shows it: the mDBC pass on the state at the start of every step of the call.
            if exc.status == ERR_DOMAIN:
                # A blob that flies apart: the engine's cell list is a DENSE grid over the bounding box of the cloud (the reference
                # sorts cell indices and has no such limit), and 2³⁰ cells is where it refuses — with a text that says so.  Legitimate
                # iff the oracle's cloud spans that many cells by the end of the call.
                x = orc.download(("Position",))["Position"]
                span = np.floor(x.max(0) / s.SimKernel.H) - np.floor(x.min(0) / s.SimKernel.H) + 3
                # (+ 4: a handle that rebuilds on the device keeps two empty cell layers on either side of the box, kStickySlack — generation 199000 met the limit within 1 %)
                assert np.prod(span + 4) > 2.0 ** 30 and "max_cells" in str(exc), f"{what}: {exc} (oracle spans {span})"
                return
                # … or if the HALF-STEP density ρₙ⁺ of some particle was non-positive in one of the steps (a violent cloud does that: the reference feeds
                # it to Pressure! and the second NeighborLoop! and the full step may well end positive again).  The oracle keeps ρₙ⁺ of its last step.
                # (Found by generation 172000: a sheet with LaminarSPS and a k = 1.41 kernel — the same refusal in round 4's tree.)
            assert exc.status == ERR_NUMERIC, f"{what}: {exc}"
            seen = False
            for k in range(done, done + steps):
                o1 = make_oracle(p, s)
                if getattr(p, "geometries", None) is not None:
                    o1.set_motions(p.geometries)
                if k:
                    o1.advance(1e9, max_steps=k)
                o2 = None
                if mdbc:
                    o1.forces_once(apply_mdbc=True)
                    seen = seen or bool((o1.download(("Density",))["Density"] <= 0).any())
                    o2 = make_oracle(p, s)
                    if getattr(p, "geometries", None) is not None:
                        o2.set_motions(p.geometries)
                    if k:
                        o2.advance(1e9, max_steps=k)
                o2 = o2 or o1
                o2.advance(1e9, max_steps=1)
                seen = seen or bool((o2.half_step_density() <= 0).any())
                # … or the density at the END of one of the steps: ρ·(2 − ε)/(2 + ε) (DensityEpsi!, :796) of a boundary particle may come out non-positive
                # and the next step's LimitDensityAtBoundary! (:794) lifts it back to ρ₀ — the oracle's state after the CALL is sane (generation 198500, same in round 4's tree)
                seen = seen or bool((o2.download(("Density",))["Density"] <= 0).any())
            assert seen, f"the engine refused a run in which neither mDBC, nor a half step, nor the end of a step produces a non-positive density in the oracle ({what}): {exc}"
            return
        done += steps
        assert (pe.iteration, pe.n_rebuilds, pe.index_counter) == (po.iteration, po.n_rebuilds, po.index_counter), what
        assert pe.total_time == pytest.approx(po.total_time, rel=1e-9), what
    e, o = _by_id(eng.download()), _by_id(orc.download())
    scale = max(np.abs(o["Position"] - o["Position"].mean(0)).max(), s.SimKernel.h)
    assert np.abs(e["Density"] - o["Density"]).max() < 1e-8 * np.abs(o["Density"]).max(), what
    assert np.abs(e["Position"] - o["Position"]).max() < 1e-9 * scale + 1e-15 * np.abs(o["Position"]).max(), what
    assert np.abs(e["Velocity"] - o["Velocity"]).max() < 1e-8 * max(np.abs(o["Velocity"]).max(), 1e-300), what
    if s.SimMetaData.KMode is StoreKernelOutput:
        (k1, g1), (k2, g2) = eng.kernel_output(), orc.kernel_output()
        i1, i2 = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
        assert np.abs(k1[i1] - k2[i2]).max() <= 1e-9 * max(np.abs(k2).max(), 1e-300), what
        assert np.abs(g1[i1] - g2[i2]).max() <= 1e-8 * max(np.abs(g2).max(), 1e-300), what
    if len(p) >= 200 and shape in ("box", "sheet", "line", "clusters"):
        # the same switches on slabs of ONE handle (two or three, along the longest axis): mDBC ghost nodes in a neighbour's slab,
        # the moving group across a cut, shifting and the kernel sums of particles whose neighbours are ghost rows
        nd = 2 + seed % 2
        try:
            # (a third of the cases cut along a chosen axis instead of the longest one)
            dd = make_engine(p, s, device_float_bytes=8, devices=[0] * nd, slab_axis=(seed // 3) % s.SimMetaData.Dimensions if seed % 3 == 0 else None)
        except Exception as exc:                        # too few cell columns for the slabs: a planning error, with a text
            assert "slab" in str(exc) or "devices" in str(exc) or "columns" in str(exc), exc
            return
        if getattr(p, "geometries", None) is not None:
            dd.set_motions(p.geometries)
        for steps in (2, 3):
            pd = dd.advance(1e9, max_steps=steps)
        assert (pd.iteration, pd.n_rebuilds, pd.index_counter) == (po.iteration, po.n_rebuilds, po.index_counter), what
        np.testing.assert_array_equal(dd.download(("ID",))["ID"], eng.download(("ID",))["ID"], err_msg=what)
        d = _by_id(dd.download())
        assert np.abs(d["Density"] - o["Density"]).max() < 1e-8 * np.abs(o["Density"]).max(), what
        assert np.abs(d["Position"] - o["Position"]).max() < 1e-9 * scale + 1e-15 * np.abs(o["Position"]).max(), what
        assert np.abs(d["Velocity"] - o["Velocity"]).max() < 1e-8 * max(np.abs(o["Velocity"]).max(), 1e-300), what
        if s.SimMetaData.KMode is StoreKernelOutput:
            k3, g3 = dd.kernel_output()
            i3 = np.argsort(dd.download(("ID",))["ID"], kind="stable")
            assert np.abs(k3[i3] - k2[i2]).max() <= 1e-9 * max(np.abs(k2).max(), 1e-300), what
            assert np.abs(g3[i3] - g2[i2]).max() <= 1e-8 * max(np.abs(g2).max(), 1e-300), what
