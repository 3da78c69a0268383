"""BASELINE config 3 names "LDS cell-tile staging on".  The shipped kernel gathers its neighbours from L1 through per-lane queues
of accept masks instead (profiles/HISTORY.md §4.5: 0.506 against 1.025 ms per launch); the staging design is kept as an ABLATION build of the
same kernel source, `-DSPHMI_LDS_STAGE=1` (sphmi_kernels.h), which `__graft_entry__.build()` compiles into
build/variants/libsphmi_ldsstage.so.  These tests hold that build to the same parity bar as the shipped library — the measurement
in profiles/r03_lds_stage_ablation.md compares two CORRECT kernels."""
import os

import numpy as np
import pytest

from conftest import perturbed
from sphexample_amd import build

_VARIANT = build.variant_path("ldsstage")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(_VARIANT), reason="no -DSPHMI_LDS_STAGE=1 build under build/variants/ (__graft_entry__.build() makes it)")]


@pytest.fixture(autouse=True)
def _variant_library(monkeypatch):
    """Load the ablation build for the tests of this module, the shipped library again afterwards."""
    from sphexample_amd import engine
    monkeypatch.setenv("SPHMI_LIB", _VARIANT)
    engine._reset_library_cache()
    yield
    monkeypatch.delenv("SPHMI_LIB")
    engine._reset_library_cache()


def _by_id(st, key):
    return st[key][np.argsort(st["ID"], kind="stable")]


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("case,fb,tol_f,tol_s", [("dam_break_2d", 8, 1e-10, 1e-9), ("dam_break_2d", 4, 2e-4, 1e-5),
                                                   ("dam_break_3d_shipped", 8, 1e-10, 1e-9), ("dam_break_3d_shipped", 4, 2e-4, 1e-5),
                                                   ("still_wedge", 8, 1e-10, 1e-6)])
def test_staged_kernel_matches_the_oracle(request, case, fb, tol_f, tol_s):
    """One force evaluation and 20 steps from a perturbed state, against the fp64 oracle: the tolerances of tests/test_engine_gpu.py."""
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    p = perturbed(p, seed=11)
    eng, orc = make_engine(p, s, device_float_bytes=fb), make_oracle(p, s)
    (d1, a1), (d2, a2) = eng.forces_once(), orc.forces_once()
    ie, io = np.argsort(eng.download(("ID",))["ID"], kind="stable"), np.argsort(orc.download(("ID",))["ID"], kind="stable")
    assert _rel(d1[ie], d2[io]) < tol_f and _rel(a1[ie], a2[io]) < tol_f
    eng.close(); orc.close()
    eng, orc = make_engine(p, s, device_float_bytes=fb), make_oracle(p, s)
    pe, po = eng.advance(1e9, max_steps=20), orc.advance(1e9, max_steps=20)
    assert (pe.iteration, pe.n_rebuilds) == (po.iteration, po.n_rebuilds)
    e, o = eng.download(), orc.download()
    assert _rel(_by_id(e, "Density"), _by_id(o, "Density")) < tol_s
    assert _rel(_by_id(e, "Position"), _by_id(o, "Position")) < max(tol_s, 1e-11)
    eng.close(); orc.close()
