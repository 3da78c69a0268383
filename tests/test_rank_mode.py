"""sphmi_create_rank with MORE THAN ONE process — the launch shape of `torchrun bench.py --gpus N`.

RCCL refuses two ranks on one device and the test box has one GPU, so the processes here share GPU 0 and reach each other
through the shared-memory transport (SPHMI_TRANSPORT=shm, csrc/sphmi_shm.h).  Everything else is the rank-mode driver an
8-GPU run uses: every process plans the slabs from the full particle set and keeps its own, counts and index lists are
negotiated between processes, cancelled steps still post matching messages, the rebuild is collective, the per-step
reductions are MAX-allreduced bit patterns.  The union of what the ranks own must be the one-device result.

CPU part: the transport itself (barrier, SUM / MAX allreduce, neighbour messages longer than a ring) in 2–4 processes.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "rank_worker.py")


def _spawn(world, args_of, timeout=300, extra_env=None):
    uid = os.urandom(128).hex()
    env = dict(os.environ, SPHMI_TRANSPORT="shm", SPHMI_SHM_TIMEOUT="60", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, args_of(r)[0], uid, str(r), str(world)] + [str(a) for a in args_of(r)[1:]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    out = []
    try:
        for pr in procs:
            o, e = pr.communicate(timeout=timeout)
            out.append((pr.returncode, o, e))
    finally:
        for pr in procs:                      # the exact processes this test started
            if pr.poll() is None:
                pr.kill()
    return out


@pytest.mark.parametrize("world,nbytes", [(2, 1000), (3, 5 << 20), (4, 0)])
def test_shm_transport_selftest(world, nbytes):
    """Barrier + SUM/MAX allreduce of vectors longer than a reduce slot + two rounds of neighbour messages (5 MiB is
    longer than the 4 MiB ring: sends and receives have to progress together) between `world` CPU processes."""
    for rc, o, e in _spawn(world, lambda r: ("selftest", nbytes), timeout=120):
        assert rc == 0, e[-2000:]


def test_shm_transport_reports_a_missing_peer():
    """A rank whose peers never arrive fails with an error after the deadline instead of hanging."""
    uid = os.urandom(128).hex()
    env = dict(os.environ, SPHMI_TRANSPORT="shm", SPHMI_SHM_TIMEOUT="2")
    pr = subprocess.run([sys.executable, WORKER, "selftest", uid, "0", "2", "16"], env=env, capture_output=True, text=True, timeout=60)
    assert pr.returncode != 0 and "peer" in pr.stderr


def _by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,steps,fb,tol,calls,axis", [
    ("dam_break_3d_shipped", 2, 40, 8, 1e-9, 1, -1), ("dam_break_3d_shipped", 2, 40, 4, 1e-5, 1, -1),
    ("dam_break_3d_shipped", 3, 40, 8, 1e-9, 2, -1), ("dam_break_3d_shipped", 4, 30, 8, 1e-9, 1, 0),
    ("dam_break_2d", 2, 60, 8, 1e-9, 1, -1),
    ("moving_square", 2, 150, 8, 1e-9, 1, 1),          # particles cross the cut one by one: order tags through migration records
    ("dam_break_2d_mdbc", 2, 40, 8, 1e-9, 1, -1)])     # wide ghost layers, mDBC on ghost copies, serial pass 1
def test_rank_mode_processes_match_one_device(case, world, steps, fb, tol, calls, axis, request, tmp_path):
    from sphexample_amd.engine import make_engine
    res = _spawn(world, lambda r: ("run", case, steps, fb, str(tmp_path), calls, axis))
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, s = request.getfixturevalue(case)
    ref = make_engine(p, s, device_float_bytes=fb)
    progs = []
    for _ in range(calls):
        pr = ref.advance(1e9, max_steps=steps // calls)
        progs.append([pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter, pr.total_time, pr.last_dt])
    progs = np.array(progs, dtype=np.float64)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for q in parts:
        # every rank reports the loop counters of the WHOLE run: same dt sequence, rebuild cadence, occupied cells
        np.testing.assert_array_equal(q["prog"][:, :4], progs[:, :4])
        np.testing.assert_allclose(q["prog"][:, 4:], progs[:, 4:], rtol=1e-12 if fb == 8 else 1e-5)
        assert tuple(q["info"][:3]) == (world, 1, 2)                    # one local slab, shm transport
    ids = np.concatenate([q["ID"] for q in parts])
    assert len(ids) == len(p) and len(np.unique(ids)) == len(p)        # every particle owned exactly once
    got = _by_id({k: np.concatenate([q[k] for q in parts]) for k in ("ID", "Density", "Position", "Velocity")})
    r = _by_id(ref.download(("ID", "Density", "Position", "Velocity")))
    assert np.abs(got["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < tol
    assert np.abs(got["Position"] - r["Position"]).max() / np.abs(r["Position"]).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("case,world,steps,fb,calls", [("dam_break_3d_shipped", 2, 60, 8, 2), ("dam_break_3d_shipped", 4, 40, 4, 1),
                                                       ("moving_square", 2, 120, 8, 1), ("dam_break_2d_mdbc", 3, 40, 8, 1)])
def test_mailbox_exchange_is_the_allreduce_bit_for_bit(case, world, steps, fb, calls, tmp_path):
    """$SPHMI_EXCHANGE=mailbox (round-4 review, next-4): the four per-step maxima through mailboxes in device memory — posted by the control launch itself into
    every peer's box (hipIpc between the rank processes), polled in its own — instead of the transport's collective (ncclAllReduce; here, on one GPU, the
    host round trip of the shared-memory transport).  The same four words travel and the reduction is an integer maximum: every rank must end with the SAME
    bits as under the default exchange — loop counters, clock, density, position — through rebuilds (a rebuild request cancels the rest of a batch: the
    cancelled steps still post), several advance calls and moving bodies."""
    runs = {}
    for mode in ("allreduce", "mailbox"):
        out = tmp_path / mode
        out.mkdir()
        res = _spawn(world, lambda r: ("run", case, steps, fb, str(out), calls, -1), extra_env={"SPHMI_EXCHANGE": mode, "SPHMI_MBOX_TIMEOUT": "30"})
        for rc, o, e in res:
            assert rc == 0, e[-3000:]
        runs[mode] = [np.load(out / f"rank{r}.npz") for r in range(world)]
    for a, b in zip(runs["allreduce"], runs["mailbox"]):
        assert int(a["info"][6]) == 0 and int(b["info"][6]) == 1           # the second run DID go through the mailboxes
        for k in ("prog", "ID", "Density", "Position", "Velocity"):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.gpu
def test_mailbox_exchange_survives_a_second_upload(request, tmp_path):
    """Round-5 advice: `sphmi_upload` restarts the step parity while the mailbox sequence number keeps counting, so after an ODD number of steps two
    consecutive posts shared one half of a box (only host latency kept that safe).  The half now follows the sequence number: seven steps, the same
    particle set uploaded again, forty more — three ranks, mailbox exchange — against the one-device handle doing the same."""
    from sphexample_amd.engine import make_engine
    case, world, steps, odd = "dam_break_3d_shipped", 3, 40, 7
    res = _spawn(world, lambda r: ("run", case, steps, 8, str(tmp_path), 1, -1), extra_env={"SPHMI_EXCHANGE": "mailbox", "SPHMI_MBOX_TIMEOUT": "30", "SPHMI_TEST_REUPLOAD": str(odd)})
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, s = request.getfixturevalue(case)
    ref = make_engine(p, s, device_float_bytes=8)
    ref.advance(1e9, max_steps=odd)
    ref.upload_particles(p)
    pr = ref.advance(1e9, max_steps=steps)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for q in parts:
        assert int(q["info"][6]) == 1
        np.testing.assert_array_equal(q["prog"][0, 1:4], [pr.steps_done, pr.n_rebuilds, pr.index_counter])
        np.testing.assert_allclose(q["prog"][0, 5], pr.last_dt, rtol=1e-12)
    got = _by_id({k: np.concatenate([q[k] for q in parts]) for k in ("ID", "Density", "Position")})
    r = _by_id(ref.download(("ID", "Density", "Position")))
    np.testing.assert_array_equal(got["ID"], r["ID"])
    assert np.abs(got["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < 1e-9
    assert np.abs(got["Position"] - r["Position"]).max() / np.abs(r["Position"]).max() < 1e-9
