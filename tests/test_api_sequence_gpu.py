"""Random CALL SEQUENCES on the C ABI: a one-device handle and a multi-slab handle of the same case must stay the same simulation
whatever order a caller uses the boundary in.

The parity suites drive upload → advance → download.  A host program — the reference's `RunSimulation` with user callbacks
(/root/reference/src/SPHCellList.jl:881-929) — may upload again, ask for the forces between two intervals, download twice, read the sort's
permutation, set the clock, advance by time or by step count.  Round 6 found a real bug this way (a second `sphmi_upload` on a multi-slab handle
died in the collective rebuild), so the sequences are drawn at random here: every seed plays ten operations on both handles and compares loop
counters (exactly) and state (fp64 kernels: 1e-9 relative; same IDs in the same order) after each one.  `$SPHMI_SEQ_SEED0` draws fresh sequences."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED0 = int(os.environ.get("SPHMI_SEQ_SEED0", "0"))


def _same(a, b, what):
    da, db = a.download(("ID", "Density", "Position", "Velocity", "Cells")), b.download(("ID", "Density", "Position", "Velocity", "Cells"))
    np.testing.assert_array_equal(da["ID"], db["ID"], err_msg=what)
    np.testing.assert_array_equal(da["Cells"], db["Cells"], err_msg=what)
    for k in ("Density", "Position"):
        assert np.abs(da[k] - db[k]).max() <= 1e-9 * max(np.abs(da[k]).max(), 1e-300), (what, k)
    vmax = max(np.abs(da["Velocity"]).max(), 1e-9)
    assert np.abs(da["Velocity"] - db["Velocity"]).max() <= 1e-7 * vmax, (what, "Velocity")


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + 12))
def test_random_call_sequences_keep_slabs_and_one_device_together(seed, request):
    from sphexample_amd.engine import make_engine
    rng = np.random.default_rng(1000 + seed)
    case = ["dam_break_2d", "dam_break_3d_shipped", "moving_square", "dam_break_2d_mdbc"][seed % 4]
    world = int(rng.choice([2, 3]))
    p, s = request.getfixturevalue(case)
    one = make_engine(p, s, device_float_bytes=8)
    dd = make_engine(p, s, device_float_bytes=8, devices=[0] * world)
    if os.environ.get("SPHMI_EXPECT_TRANSPORT"):               # (test_the_same_sequences_over_the_rccl_branch below: the slabs must really talk through it)
        assert dd.multi_info().transport == int(os.environ["SPHMI_EXPECT_TRANSPORT"])
    log = [f"{case} x{world}"]
    perm_one = perm_dd = None
    for step in range(10):
        op = rng.choice(["advance_steps", "advance_steps", "advance_time", "forces_once", "download_twice", "permutation", "reupload", "set_clock"])
        log.append(str(op))
        what = " → ".join(log)
        if op == "advance_steps":
            k = int(rng.integers(1, 26))
            pa, pb = one.advance(1e9, max_steps=k), dd.advance(1e9, max_steps=k)
        elif op == "advance_time":
            pa0 = one.advance(1e9, max_steps=0)                       # (a call that takes no step: reads the clock)
            t = pa0.total_time + float(rng.uniform(0.5, 6.0)) * max(pa0.last_dt, 2e-5)
            pa, pb = one.advance(t), dd.advance(t)
        elif op == "forces_once":
            (d1, a1), (d2, a2) = one.forces_once(apply_mdbc=case.endswith("mdbc")), dd.forces_once(apply_mdbc=case.endswith("mdbc"))
            assert np.abs(d1 - d2).max() <= 1e-9 * max(np.abs(d1).max(), 1e-300), what
            assert np.abs(a1 - a2).max() <= 1e-9 * max(np.abs(a1).max(), 1e-300), what
            continue
        elif op == "download_twice":
            _same(one, dd, what); _same(one, dd, what + " (again)")
            continue
        elif op == "permutation":
            perm_one, perm_dd = one.download_permutation(), dd.download_permutation()
            np.testing.assert_array_equal(perm_one, perm_dd, err_msg=what)
            assert sorted(perm_one.tolist()) == list(range(len(p))), what
            continue
        elif op == "reupload":
            st = one.download()
            q = p.copy()
            o = np.argsort(st["ID"], kind="stable")                    # the CURRENT state, back in ID order, goes up again on both
            for f in ("Position", "Velocity", "Acceleration", "Density"):
                getattr(q, f)[...] = st[f][o]
            if getattr(p, "geometries", None) is not None:
                q.geometries = p.geometries
            one.upload_particles(q); dd.upload_particles(q)
            continue
        else:
            it, t = int(rng.integers(0, 1000)), float(rng.uniform(0.0, 0.3))
            one.set_clock(it, t); dd.set_clock(it, t)
            continue
        assert (pa.iteration, pa.steps_done, pa.n_rebuilds, pa.index_counter) == (pb.iteration, pb.steps_done, pb.n_rebuilds, pb.index_counter), what
        assert pa.total_time == pytest.approx(pb.total_time, rel=1e-12) and pa.last_dt == pytest.approx(pb.last_dt, rel=1e-12), what
        _same(one, dd, what)


@pytest.mark.parametrize("timing", ["complete_in_call", "stream_ordered"])
def test_the_same_sequences_over_the_rccl_branch(timing):
    """The sequences once more with the slabs of the handle talking through the RCCL branch (ncclCommInitAll, grouped ncclSend / ncclRecv, the per-step
    allreduce) — behind the checking double of tests/mock_rccl/, which turns a send without its receive, a size mismatch or an open group into an error:
    a second upload, forces_once between two intervals and the rest must leave both ends of every exchange in step.  A subprocess: the library binds
    its RCCL once per process.  `stream_ordered`: the double in its asynchronous mode ($MOCK_RCCL_ASYNC=1, calls only queue; fresh sequences) — a
    download, a second upload or forces_once that did not wait for an exchange still in flight would read or overwrite its buffers."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "mock_rccl"))
    try:
        import build as mock_build
        mock = mock_build.build()
    finally:
        sys.path.remove(os.path.join(here, "mock_rccl"))
    env = dict(os.environ, SPHMI_TRANSPORT="rccl", SPHMI_RCCL_LIB=mock, SPHMI_EXPECT_TRANSPORT="1", SPHMI_SEQ_SEED0=str(SEED0 + 5000), MOCK_RCCL_TIMEOUT="60")
    if timing == "stream_ordered":
        env.update(MOCK_RCCL_ASYNC="1", MOCK_RCCL_ASYNC_DELAY_US="500", GPU_MAX_HW_QUEUES="24", SPHMI_SEQ_SEED0=str(SEED0 + 6000))
    pr = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "random_call_sequences"],
                        env=env, capture_output=True, text=True, timeout=1200)
    assert pr.returncode == 0, pr.stdout[-3000:] + pr.stderr[-2000:]
    assert "12 passed" in pr.stdout and "VIOLATION" not in pr.stderr
