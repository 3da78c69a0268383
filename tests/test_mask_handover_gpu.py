"""The corrector of a step takes its accept masks from the predictor of the same step (ForceParams::mstore) instead
of running phase 1 again: plain 3-D fp32 handles whose launches run one wave per tile.  The predictor tests against
H + vmax·Δt, so the handed-over masks are a SUPERSET of the pairs the corrector meets at the half-step positions; the pairs
beyond H contribute exactly zero (the Wendland factor is clamped), and the order of a lane's pairs is the order of the chunks —
so the state must come out BIT FOR BIT the same with the hand-over on, off, and with a capacity so small that most chunks
fall back to scanning.  A missing pair (a skin that is too thin) would show as a difference."""
import os

import numpy as np
import pytest

from conftest import flowing, perturbed
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d

# The hand-over is an EXPERIMENT build (measured and off: sphmi_kernels.h): these tests run against a library built
# with -DSPHMI_MASK_STORE=1 — `python tools/prebuild_variants.py "maskstore:-DSPHMI_MASK_STORE=1"` — and skip without one.
_VARIANT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "variants", "libsphmi_maskstore.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(_VARIANT), reason="no -DSPHMI_MASK_STORE=1 build under build/variants/")]


@pytest.fixture(autouse=True)
def _variant_library(monkeypatch):
    """Load the experiment build for the tests of this module, the shipped library again afterwards."""
    from sphexample_amd import engine
    monkeypatch.setenv("SPHMI_LIB", _VARIANT)
    engine._reset_library_cache()
    yield
    monkeypatch.delenv("SPHMI_LIB")
    engine._reset_library_cache()


def _run(p, s, steps, monkeypatch, store, cap=None, calls=1):
    from sphexample_amd.engine import make_engine
    monkeypatch.setenv("SPHMI_WPT", "1")                   # one wave per tile on a case this small (the switch follows the tile count)
    monkeypatch.setenv("SPHMI_MASK_STORE", store)
    if cap is None:
        monkeypatch.delenv("SPHMI_MASK_CAP", raising=False)
    else:
        monkeypatch.setenv("SPHMI_MASK_CAP", str(cap))
    eng = make_engine(p, s, device_float_bytes=4)
    for _ in range(calls):
        pr = eng.advance(1e9, max_steps=steps)
    st = eng.download()
    return pr, st


def _same(a, b):
    for k in ("ID", "Position", "Velocity", "Density", "Pressure", "Acceleration"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("state", ["rest", "perturbed", "flowing", "fast"])
def test_hand_over_changes_nothing(dam_break_3d_shipped, state, monkeypatch):
    p, s = dam_break_3d_shipped
    if state == "perturbed":
        p = perturbed(p, seed=4, vel_scale=0.5)
    elif state == "flowing":
        p = flowing(p, seed=5, shear=2.0, base=1.0)
    elif state == "fast":                                   # Mach ≈ 0.3: a skin of several per cent of H, rebuilds every few steps
        p = perturbed(p, seed=6, vel_scale=10.0)
    steps = 40 if state != "fast" else 12
    ref_pr, ref = _run(p, s, steps, monkeypatch, "0")
    for cap in (None, 5, 1):
        pr, st = _run(p, s, steps, monkeypatch, "1", cap)
        assert (pr.iteration, pr.n_rebuilds, pr.total_time, pr.last_dt) == (ref_pr.iteration, ref_pr.n_rebuilds, ref_pr.total_time, ref_pr.last_dt)
        _same(st, ref)


def test_hand_over_across_calls_and_rebuilds(dam_break_3d_shipped, monkeypatch):
    """Short calls (every sphmi_advance starts with a rebuild and a cancelled first control), odd and even step counts: the slot
    that carries max |v|² flips with the reduction slots."""
    p, s = dam_break_3d_shipped
    p = flowing(p, seed=8, shear=1.5, base=0.8)
    from sphexample_amd.engine import make_engine

    def run(store):
        monkeypatch.setenv("SPHMI_WPT", "1")
        monkeypatch.setenv("SPHMI_MASK_STORE", store)
        eng = make_engine(p, s, device_float_bytes=4)
        out = []
        for n in (1, 2, 3, 1, 7, 4, 11):
            pr = eng.advance(1e9, max_steps=n)
            out.append((pr.iteration, pr.n_rebuilds, pr.total_time))
        return out, eng.download()
    o0, s0 = run("0")
    o1, s1 = run("1")
    assert o0 == o1
    _same(s1, s0)


def test_hand_over_against_the_oracle_at_a_size_that_uses_it_by_default(monkeypatch):
    """≈0.4 M particles = 6.3 k tiles: the launches run one wave per tile on their own and the hand-over is on by default;
    fp32 engine against the fp64 oracle over 12 steps of a moving state (1e-5), and against the engine with the hand-over off
    (bit for bit)."""
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    dp = 0.0059
    p, s = flowing(dam_break_3d(dp), seed=2, shear=1.0, base=0.5), setup_dam_break_3d(dp)
    assert len(p) > 6000 * 64
    orc = make_oracle(p, s)
    po = orc.advance(1e9, max_steps=12)
    o = orc.download()
    io = np.argsort(o["ID"])
    res = {}
    for store in ("1", "0"):
        monkeypatch.setenv("SPHMI_MASK_STORE", store)
        eng = make_engine(p, s, device_float_bytes=4)
        pe = eng.advance(1e9, max_steps=12)
        assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
        e = eng.download()
        ie = np.argsort(e["ID"])
        assert np.abs(e["Density"][ie] - o["Density"][io]).max() / np.abs(o["Density"]).max() < 1e-5
        assert np.abs(e["Position"][ie] - o["Position"][io]).max() / np.abs(o["Position"]).max() < 1e-5
        res[store] = e
    _same(res["1"], res["0"])
