"""The stock examples in the precision a user GETS (round-4 review, "Next round" 1).

The Julia shim and `make_engine` leave the arithmetic to the library (`device_float_bytes = 0`, `sphmi_auto_device_float_bytes`):
fp32 kernels with double-float state where every term of the path is continuous in the positions (the kernel vanishes at its
cut-off, k >= 2, and no mDBC: the dam breaks), fp64 kernels where the reference's algorithm has a discontinuity that an fp32
trajectory takes a step early or late —
  * k < 2 (DucklingMDBC 1.5, MovingSquare2d sqrt 2): a pair at r ~ H switches a finite force, fp32 rounding of r² against H² decides
    thousands of lattice ties differently from fp64, and the fp32 state leaves the oracle's by 1e-5 within ~150 / 50 steps;
  * mDBC (src/SPHCellList.jl:598-622): "no neighbour -> keep the density", rho := b1/A11 for a lone neighbour at r ~ H (a 0/0),
    |det A| >= 1e-3 — an fp32 trajectory freezes 1e-4-level differences into single boundary particles of a streaming Dambreak2dMDBC.
profiles/r05_fp32_examples_parity.md holds the figures of every layout in both precisions.

Every one of the five mDBC / moving-body layouts runs >= 100 steps in its DEFAULT precision against the fp64 oracle
(oracle/sph_oracle.c restating /root/reference/src/SPHCellList.jl:219-266,575-622,727-805), as shipped ("rest") and in a streaming
state whose Δx criterion (:758-762) asks for >= 2 rebuilds inside the window: same rebuild steps, IndexCounter and clock, ρ and x
1e-9 relative.  The three k = 2 mDBC layouts ALSO run forced to fp32 (what $SPHMI_DEVICE_FLOAT_BYTES=4 gives): every fluid particle's
ρ and every x < 1e-5 (north_star), boundary ρ < 1e-5 but for a handful of particles behind the branches above.
"""
import numpy as np
import pytest

import conftest
from conftest import flowing

pytestmark = pytest.mark.gpu

# case → (arithmetic the policy must choose, steps, streaming state or None: the Moving square forces the rebuilds itself)
EXAMPLES = {
    "still_wedge": (8, 100, dict(seed=5, base=0.4, shear=0.4, noise=0.0, rho_scale=0.0)),
    "dam_break_2d_mdbc": (8, 150, dict(seed=5, base=1.0, shear=1.0, noise=0.0, rho_scale=0.3)),
    "still_wedge_middle_square": (8, 100, dict(seed=5, base=0.4, shear=0.4, noise=0.0, rho_scale=0.0)),
    "duckling": (8, 100, dict(seed=5, base=0.8, shear=0.8, noise=0.02, rho_scale=0.3)),
    "moving_square": (8, 100, None),
}
TOL = {4: (1e-5, 1e-5, 1e-5), 8: (1e-9, 1e-9, 1e-9)}        # ρ, x (relative to the field maximum), Δt and clock (relative)


def _by_id(st):
    order = np.argsort(st["ID"], kind="stable")
    return {k: v[order] for k, v in st.items()}


def _run(case, state, fb=0):
    from oracle.oracle import Oracle, make_oracle
    from sphexample_amd.engine import make_engine
    p0, s = getattr(conftest, "load_" + case)()
    want, steps, flow = EXAMPLES[case]
    p = p0
    if state == "flow":
        p = flowing(p0, **flow)
        if hasattr(p0, "geometries"): p.geometries = p0.geometries
    eng, orc = make_engine(p, s, device_float_bytes=fb), make_oracle(p, s, threads=min(8, Oracle.max_threads()))
    if hasattr(p0, "geometries"):
        orc.set_motions(p0.geometries)
    out = []
    for n in (steps // 2, steps - steps // 2):
        pe, po = eng.advance(1e9, max_steps=n), orc.advance(1e9, max_steps=n)
        e, o = _by_id(eng.download()), _by_id(orc.download())
        rho_all = np.abs(e["Density"] - o["Density"]) / np.abs(o["Density"]).max()
        out.append(dict(pe=pe, po=po, rho=rho_all.max(), rho_fluid=rho_all[o["Type"] == 1].max(), rho_bnd=rho_all[o["Type"] != 1],
                        x=np.abs(e["Position"] - o["Position"]).max() / np.abs(o["Position"]).max(),
                        dt=abs(pe.last_dt - po.last_dt) / po.last_dt, t=abs(pe.total_time - po.total_time) / po.total_time))
    return eng, out


@pytest.mark.parametrize("case", list(EXAMPLES))
def test_the_library_chooses_the_arithmetic_of_every_stock_example(case):
    """`device_float_bytes = 0`: fp32 for H >= 2h, fp64 for kernels cut off before they vanish."""
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + case)()
    assert make_engine(p, s).device_float_bytes == EXAMPLES[case][0]
    assert make_engine(p, s, device_float_bytes=8).device_float_bytes == 8 and make_engine(p, s, device_float_bytes=4).device_float_bytes == 4


@pytest.mark.parametrize("state", ["rest", "flow"])
@pytest.mark.parametrize("case", list(EXAMPLES))
def test_example_layouts_in_their_default_precision_track_the_oracle(case, state):
    want, steps, flow = EXAMPLES[case]
    if state == "flow" and flow is None:
        pytest.skip("the Moving body forces the rebuilds in the layout as shipped")
    eng, out = _run(case, state)
    assert eng.device_float_bytes == want
    tol_rho, tol_x, tol_t = TOL[want]
    for r in out:
        pe, po = r["pe"], r["po"]
        assert pe.iteration == po.iteration and (pe.n_rebuilds, pe.index_counter) == (po.n_rebuilds, po.index_counter), (case, state, pe.n_rebuilds, po.n_rebuilds)
        assert r["rho"] < tol_rho and r["x"] < tol_x and r["dt"] < tol_t and r["t"] < tol_t, (case, state, {k: v for k, v in r.items() if k in ("rho", "x", "dt", "t")})
    assert out[-1]["pe"].iteration == steps
    if state == "flow" or case == "moving_square":
        # the two rebuilds that open the two calls + at least two that the Δx criterion asked for inside the window
        assert out[-1]["pe"].n_rebuilds >= 4, out[-1]["pe"].n_rebuilds


@pytest.mark.parametrize("state", ["rest", "flow"])
@pytest.mark.parametrize("case", ["still_wedge", "dam_break_2d_mdbc", "still_wedge_middle_square"])
def test_mdbc_layouts_forced_to_fp32(case, state):
    """What $SPHMI_DEVICE_FLOAT_BYTES=4 gives on the k = 2 mDBC examples, >= 100 steps with rebuilds against the fp64 oracle: the fluid and
    every position hold the north-star 1e-5; the boundary does too but for the particles whose ghost node sits on one of mDBC's branches
    (a lone neighbour at r ~ H, no neighbour at all: src/SPHCellList.jl:598-622) — at most eight of them, each below 2e-3.
    As shipped, the three layouts hold 3e-7 everywhere over 200 steps (profiles/r05_fp32_examples_parity.md)."""
    want, steps, flow = EXAMPLES[case]
    eng, out = _run(case, state, fb=4)
    assert eng.device_float_bytes == 4
    for r in out:
        pe, po = r["pe"], r["po"]
        assert pe.iteration == po.iteration and pe.n_rebuilds == po.n_rebuilds
        # (occupied cells: fp32 handles hash fp32 positions, and Dambreak2dMDBC.jl's lattice puts particles ON cell faces — 1 841 cells against 1 839)
        assert abs(pe.index_counter - po.index_counter) <= 4
        fig = {k: v for k, v in r.items() if k in ("rho_fluid", "x", "dt", "t")}
        assert r["rho_fluid"] < 1e-5 and r["x"] < 1e-5 and r["dt"] < 1e-5 and r["t"] < 1e-5, (case, state, fig)
        bnd = r["rho_bnd"]
        assert (bnd > 1e-5).sum() <= 8 and bnd.max() < 2e-3, (case, state, int((bnd > 1e-5).sum()), float(bnd.max()))
        if state == "rest":
            assert bnd.max() < 1e-5, (case, float(bnd.max()))
    if state == "flow":
        assert out[-1]["pe"].n_rebuilds >= 4


@pytest.mark.parametrize("case,steps,floor", [("duckling", 200, 1e-5), ("moving_square", 50, 1e-4)])
def test_fp32_on_kernels_cut_off_before_they_vanish_is_why_the_policy_exists(case, steps, floor):
    """The record behind the policy, kept honest: forced fp32 handles on the two k < 2 examples DO leave the oracle by more than the
    north-star tolerance within this window (DucklingMDBC 2.9e-5 after 200 steps, MovingSquare2d 7e-3 after 50).  If a future kernel
    holds 1e-5 here, this test fails and the policy should be revisited."""
    from oracle.oracle import Oracle, make_oracle
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + case)()
    eng, orc = make_engine(p, s, device_float_bytes=4), make_oracle(p, s, threads=min(8, Oracle.max_threads()))
    if hasattr(p, "geometries"):
        orc.set_motions(p.geometries)
    eng.advance(1e9, max_steps=steps); orc.advance(1e9, max_steps=steps)
    e, o = _by_id(eng.download()), _by_id(orc.download())
    err = np.abs(e["Density"] - o["Density"]).max() / np.abs(o["Density"]).max()
    assert err > floor, err
    assert err < 0.1, err          # … while staying a sane simulation
