"""sphmi_download_permutation — the sort as a permutation.

The reference's sort! permutes all 17 fields of the SimParticles StructArray (src/SPHCellList.jl:142); the engine carries
ten of them.  prev_row lets the caller bring the rest along with one gather per field instead of sorting by ID on the host.
CPU part: the oracle's own row-number column (one more column its sort permutes).  GPU part: the engine's permutation is
the oracle's, interval by interval, on one device and on slabs; RunSimulation keeps GhostNormals / ChunkID / user-visible
passive fields consistent with the oracle-backed driver.
"""
import copy

import numpy as np
import pytest


def _intervals(backend, n_calls, steps):
    """(ID column, prev_row) after each of n_calls advances."""
    out = []
    for _ in range(n_calls):
        backend.advance(1e9, max_steps=steps)
        ids = backend.download(("ID",))["ID"]
        out.append((ids, backend.download_permutation()))
    return out


def test_oracle_permutation_is_the_row_history(dam_break_2d):
    from conftest import perturbed
    from oracle.oracle import make_oracle
    p, s = dam_break_2d
    q = perturbed(p, seed=1, vel_scale=3.0)                # fast enough for Δx-triggered re-sorts inside the window
    orc = make_oracle(q, s)
    prev_ids = q.ID.copy()
    tag = np.arange(len(q)) * 10 + 7                       # a passive user column riding along
    moved = 0
    for ids, prow in _intervals(orc, 4, 25):
        assert sorted(prow) == list(range(len(q)))         # a permutation
        np.testing.assert_array_equal(ids, prev_ids[prow]) # row i now was row prow[i] before
        tag = tag[prow]
        moved += int((prow != np.arange(len(q))).sum())
        prev_ids = ids
    # composing the interval permutations = looking the final IDs up in the initial order
    first_row = {int(v): k for k, v in enumerate(q.ID)}
    np.testing.assert_array_equal((tag - 7) // 10, [first_row[int(v)] for v in prev_ids])
    assert moved > 0
    # a second call without an advance in between: identity
    np.testing.assert_array_equal(orc.download_permutation(), np.arange(len(q)))


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
def test_engine_permutation_equals_the_oracles(dam_break_2d, devices):
    """fp64 kernels keep the oracle's order ID for ID, so the permutations must be equal interval by interval — one device
    (the 4-byte row column that travels through the engine's sort) and multi-device handles (the same column, riding in the migration and ghost-layer records)."""
    from conftest import perturbed
    from oracle.oracle import make_oracle
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    q = perturbed(p, seed=1, vel_scale=3.0)
    eng = make_engine(q, s, device_float_bytes=8, devices=devices)
    orc = make_oracle(q, s)
    for (ie, pe), (io, po) in zip(_intervals(eng, 4, 25), _intervals(orc, 4, 25)):
        np.testing.assert_array_equal(ie, io)
        np.testing.assert_array_equal(pe, po)
    np.testing.assert_array_equal(eng.download_permutation(), np.arange(len(q)))
    assert eng.advance(1e9, max_steps=1).n_rebuilds >= 3


@pytest.mark.gpu
def test_run_simulation_keeps_the_passive_fields(dam_break_2d_mdbc):
    """GhostNormals, ChunkID, GravityFactor, MotionLimiter, BoundaryBool after every output interval: the engine-backed
    RunSimulation (permutation from the engine, asynchronous download) = the oracle-backed one (permutation from the
    oracle's sort), and both are consistent with the carried fields (GravityFactor follows Type, ChunkID follows ID)."""
    from oracle.oracle import Oracle
    from sphexample_amd.simulation import PASSIVE_FIELDS, RunSimulation
    p, s = dam_break_2d_mdbc
    got = {}
    for name, kw in (("gpu", dict(device_float_bytes=8, async_output=True)), ("gpu_sync", dict(device_float_bytes=8)), ("cpu", dict(backend_factory=Oracle))):
        meta = copy.deepcopy(s.SimMetaData)
        meta.SimulationTime, meta.OutputTimes = 0.004, 0.001
        q = p.copy()
        q.ChunkID[:] = q.ID * 3 + 1                         # a passive column with a known relation to a carried one
        assert np.abs(q.GhostNormals).max() > 0
        normals_of_id = {int(i): tuple(n) for i, n in zip(q.ID, q.GhostNormals)}
        snaps = []
        RunSimulation(SimGeometry=None, SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel, SimLogger=None,
                      SimParticles=q, SimViscosity=s.SimViscosity, SimDensityDiffusion=s.SimDensityDiffusion,
                      on_output=lambda m, pp: snaps.append({k: getattr(pp, k).copy() for k in PASSIVE_FIELDS + ("ID", "Type")}), **kw)
        got[name] = snaps
        for sn in snaps:
            np.testing.assert_array_equal(sn["ChunkID"], sn["ID"] * 3 + 1)
            np.testing.assert_array_equal(sn["GravityFactor"], np.where(sn["Type"] == 1, -1.0, np.where(sn["Type"] == 3, 1.0, 0.0)))
            np.testing.assert_array_equal(sn["MotionLimiter"], (sn["Type"] == 1).astype(float))
            np.testing.assert_array_equal(sn["BoundaryBool"], (sn["Type"] != 1).astype(np.uint8))
            assert all(tuple(n) == normals_of_id[int(i)] for i, n in zip(sn["ID"][::97], sn["GhostNormals"][::97]))
    assert len(got["gpu"]) == len(got["cpu"]) == len(got["gpu_sync"]) >= 5
    for a, b, c in zip(got["gpu"], got["cpu"], got["gpu_sync"]):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            np.testing.assert_array_equal(c[k], b[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
def test_permutation_does_not_depend_on_unique_ids(dam_break_2d, devices):
    """The reference reads Idp per CSV file and concatenates the geometries (src/PreProcess.jl:28,71): IDs may repeat.  The
    row column is a column of its own — on slabs it rides in the migration and ghost-layer records — so the permutation of a
    set whose IDs are ALL EQUAL is the permutation of the same set with unique IDs (round-3 advice: the slab path matched
    ID columns and returned a non-permutation without a word)."""
    from conftest import perturbed
    from sphexample_amd.engine import make_engine
    p, s = dam_break_2d
    q = perturbed(p, seed=1, vel_scale=3.0)
    dup = q.copy()
    dup.ID[:] = 7
    ref = make_engine(q, s, device_float_bytes=8, devices=devices)
    eng = make_engine(dup, s, device_float_bytes=8, devices=devices)
    passive = np.arange(len(q))
    for (_, pr), (ie, pe) in zip(_intervals(ref, 4, 25), _intervals(eng, 4, 25)):
        assert (ie == 7).all()
        np.testing.assert_array_equal(pe, pr)
        assert sorted(pe) == list(range(len(q)))
        passive = passive[pe]
    # the passive column now names the upload row of every particle: its uploaded position is where that row was
    x_now = ref.download(("Position",))["Position"]
    x_dup = eng.download(("Position",))["Position"]
    np.testing.assert_array_equal(x_now, x_dup)
    ref.close(); eng.close()
