"""Pins against the REAL reference: state the unmodified SPHExample leaves after one short output interval, produced by
tools/dump_fixture.jl on a machine with Julia ≥ 1.11 and committed under tests/golden/reference/.

The build image has no Julia, so the files cannot be produced there; until someone runs the script these tests SKIP and
the oracle stays "parity unpinned" (oracle/sph_oracle.c header).  With the files present:
  * the CPU oracle must reproduce the reference's iteration count, clock, Δt and ID-sorted state to 1e-9 relative
    (fp64 both sides; the reference's summation order depends on its thread count — SURVEY.md §8a Q8);
  * the HIP engine (fp64 kernels, -m gpu) the same.
"""
import gzip
import json
import os

import numpy as np
import pytest

import conftest

REF = os.path.join(conftest.GOLDEN, "reference")
CASES = {
    "still_wedge_mdbc": "load_still_wedge",
    "dam_break_2d_mdbc": "load_dam_break_2d_mdbc",
    "dam_break_2d": "load_dam_break_2d",
    "dam_break_3d_dp0.02": "load_dam_break_3d_shipped",
    "moving_square_2d_dp0.04": "load_moving_square",
}
TOL = 1e-9


def _fixture(case):
    meta_p, tab_p = os.path.join(REF, case + ".json"), os.path.join(REF, case + ".csv.gz")
    if not (os.path.exists(meta_p) and os.path.exists(tab_p)):
        pytest.skip(f"no reference fixture for {case}: run tools/dump_fixture.jl with Julia >= 1.11 (see tests/golden/README.md)")
    meta = json.load(open(meta_p))
    with gzip.open(tab_p, "rt") as f:
        names = f.readline().strip().split(",")
        tab = np.loadtxt(f, delimiter=",", ndmin=2)
    col = {n: tab[:, i] for i, n in enumerate(names)}
    D = int(meta["dims"])
    vec = lambda p: np.stack([col[f"{p}{d}"] for d in range(1, D + 1)], axis=1)  # noqa: E731
    return meta, {"ID": col["ID"].astype(np.int64), "Position": vec("x"), "Velocity": vec("v"), "Acceleration": vec("a"),
                  "Density": col["rho"], "Pressure": col["p"], "Type": col["type"].astype(np.int64)}


def _check(backend, case, meta, ref):
    pr = backend.advance(float(meta["t_target"]))             # one SimulationLoop call, as the script's RunSimulation made
    assert pr.iteration == meta["iteration"]
    assert pr.total_time == pytest.approx(meta["total_time"], rel=1e-12)
    assert pr.last_dt == pytest.approx(meta["last_dt"], rel=1e-10)
    assert pr.index_counter == meta["index_counter"]
    st = backend.download(("ID", "Position", "Velocity", "Acceleration", "Density", "Pressure", "Type"))
    o = np.argsort(st["ID"], kind="stable")
    np.testing.assert_array_equal(st["ID"][o], ref["ID"])
    np.testing.assert_array_equal(st["Type"][o], ref["Type"])
    for k in ("Position", "Velocity", "Acceleration", "Density", "Pressure"):
        scale = max(np.abs(ref[k]).max(), 1e-300)
        assert np.abs(st[k][o] - ref[k]).max() / scale < TOL, (case, k)


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_reproduces_the_reference(case):
    from oracle.oracle import make_oracle
    meta, ref = _fixture(case)
    p, s = getattr(conftest, CASES[case])()
    assert len(p) == meta["n"]
    orc = make_oracle(p, s)
    if hasattr(p, "geometries"):
        orc.set_motions(p.geometries)
    _check(orc, case, meta, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_engine_reproduces_the_reference(case):
    from sphexample_amd.engine import make_engine
    meta, ref = _fixture(case)
    p, s = getattr(conftest, CASES[case])()
    eng = make_engine(p, s, device_float_bytes=8)
    _check(eng, case, meta, ref)
