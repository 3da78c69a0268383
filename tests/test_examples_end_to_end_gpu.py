"""A shortened run of tools/examples_end_to_end.py in the suite (round-4 review, "Next round" 3): every stock example through the per-interval
call pattern of RunSimulation (/root/reference/src/SPHCellList.jl:881-929 — one SimulationLoop call + one download per OutputTimes
interval) in the precision the library chooses, asynchronous output against synchronous output bit for bit, no SPHMI_ERR_*, no NaN.
The full-length record (Dambreak3d.jl to 1.6 s, the others to their SimulationTime): profiles/r05_examples_end_to_end.md."""
import copy

import numpy as np
import pytest

import conftest
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.simulation import RunSimulation

pytestmark = pytest.mark.gpu

CASES = ["dam_break_3d_dx0.0085", "still_wedge", "dam_break_2d_mdbc", "still_wedge_middle_square", "duckling", "moving_square"]


def _load(name):
    if name == "dam_break_3d_dx0.0085":
        return dam_break_3d(0.0085), setup_dam_break_3d(0.0085)
    return getattr(conftest, "load_" + name)()


@pytest.mark.parametrize("case", CASES)
def test_five_output_intervals_of_every_stock_example(case):
    p0, s = _load(case)
    runs = []
    for async_output in (True, False):
        p = p0.copy()
        meta = copy.deepcopy(s.SimMetaData)
        first = meta.OutputTimes if np.isscalar(meta.OutputTimes) else meta.OutputTimes[0]
        meta.SimulationTime = 5 * first
        seen = []
        steps = RunSimulation(SimGeometry=getattr(p0, "geometries", None), SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel,
                              SimParticles=p, SimViscosity=s.SimViscosity, SimDensityDiffusion=s.SimDensityDiffusion,
                              on_output=lambda md, P: seen.append((md.Iteration, md.TotalTime, float(P.Density.sum()))), async_output=async_output)
        assert meta.TotalTime > meta.SimulationTime and len(steps) >= 5 and len(seen) == len(steps) + 1
        assert all(b[1] > a[1] for a, b in zip(seen, seen[1:]))                     # every callback saw a later snapshot
        assert not np.isnan(p.Position).any() and (p.Density > 0).all()
        runs.append((meta.Iteration, seen, p))
    (ia, sa, pa), (ib, sb, pb) = runs
    assert ia == ib and sa == sb
    oa, ob = np.argsort(pa.ID), np.argsort(pb.ID)
    assert np.array_equal(pa.Position[oa], pb.Position[ob]) and np.array_equal(pa.Density[oa], pb.Density[ob])
