"""The RCCL branch of the slab driver with MORE THAN ONE rank — on one GPU, through the checking double of tests/mock_rccl/.

RCCL refuses two ranks on one device and the test box has one GPU, so until round 6 the branch of csrc/sphmi_multi.h that a real
8-GPU run takes — ncclCommInitRank twice (the second id summed over the first communicator), the neighbour counts and migration
records as ncclSend / ncclRecv groups on the main stream, the halos as groups on the side stream, the per-step 4-word MAX-allreduce
on the second communicator, the rebuild-time host scalars, the mailbox handle table — had only ever run with world = 1; the
2-4-process tests of tests/test_rank_mode.py drive the same slab logic through the shared-memory transport, which takes a different
branch of `exchange`, `host_allreduce`, `host_neighbour_counts` and `reductions_and_control`.  Two ranks whose message lists differ
in length or order hang a real RCCL; the double (mock_rccl.cpp; its own checks are tested in tests/test_mock_rccl.py) turns that
into an error that names the ranks and the message.

Every test: `world` processes (or one process with `world` slabs: the ncclCommInitAll branch) with $SPHMI_RCCL_LIB pointing at the
double, transport == 1 (RCCL) reported by the handle, zero violations and no communicator left alive in the double's counters, and
the union of what the ranks own equal to the one-device handle: same IDs, same loop counters, state within the precision's tolerance
(fp64 1e-9, fp32 1e-5 relative on density and position — summation order inside a cell row differs between slab and whole domain)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "rank_worker.py")
MOCK_DIR = os.path.join(HERE, "mock_rccl")


@pytest.fixture(scope="module")
def mock_lib():
    sys.path.insert(0, MOCK_DIR)
    try:
        import build as mock_build
        return mock_build.build()
    finally:
        sys.path.remove(MOCK_DIR)


def _spawn(mock, world, args_of, mode="run", timeout=600, extra_env=None, n_procs=None):
    uid = os.urandom(128).hex()
    env = dict(os.environ, SPHMI_RCCL_LIB=mock, MOCK_RCCL_TIMEOUT="90")
    env.update(extra_env or {})
    if mode == "run":
        env.pop("SPHMI_TRANSPORT", None)                 # rank mode without SPHMI_TRANSPORT=shm IS the RCCL branch
    else:
        env["SPHMI_TRANSPORT"] = "rccl"                  # one process, slabs sharing a device: allowed with a substitute library only
    procs = [subprocess.Popen([sys.executable, WORKER, mode, uid, str(r), str(world)] + [str(a) for a in args_of(r)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world if n_procs is None else n_procs)]
    out = []
    try:
        for pr in procs:
            o, e = pr.communicate(timeout=timeout)
            out.append((pr.returncode, o, e))
    finally:
        for pr in procs:                                 # the exact processes this test started
            if pr.poll() is None:
                pr.kill()
    return out


def _by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


def _reference(request, case, fb, steps, calls):
    from sphexample_amd.engine import make_engine
    p, s = request.getfixturevalue(case)
    ref = make_engine(p, s, device_float_bytes=fb)
    progs = []
    for _ in range(calls):
        pr = ref.advance(1e9, max_steps=steps // calls)
        progs.append([pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter, pr.total_time, pr.last_dt])
    return p, ref, np.array(progs, dtype=np.float64)


def _check_union(parts, p, ref, progs, fb, tol, n_local):
    for q in parts:
        np.testing.assert_array_equal(q["prog"][:, :4], progs[:, :4])            # dt sequence, rebuild cadence, occupied cells: the WHOLE run's
        np.testing.assert_allclose(q["prog"][:, 4:], progs[:, 4:], rtol=1e-12 if fb == 8 else 1e-5)
        assert tuple(q["info"][:3]) == (len(parts) if n_local == 1 else n_local, n_local, 1), q["info"]     # transport 1 = RCCL
    ids = np.concatenate([q["ID"] for q in parts])
    assert len(ids) == len(p) and len(np.unique(ids)) == len(p)                    # every particle owned exactly once
    got = _by_id({k: np.concatenate([q[k] for q in parts]) for k in ("ID", "Density", "Position", "Velocity")})
    r = _by_id(ref.download(("ID", "Density", "Position", "Velocity")))
    np.testing.assert_array_equal(got["ID"], r["ID"])
    assert np.abs(got["Density"] - r["Density"]).max() / np.abs(r["Density"]).max() < tol
    assert np.abs(got["Position"] - r["Position"]).max() / np.abs(r["Position"]).max() < tol


def _check_mock(parts, world, two_comms=True, overlapped=True, proxied=False):
    for rank, q in enumerate(parts):
        m = [int(x) for x in q["mock"]]
        comms, groups, sends, recvs, sbytes, rbytes, allred, viol, p2p_streams, coll_streams, _, alive, depth, late, by_proxy = m
        assert (by_proxy == groups and groups > 0) if proxied else by_proxy == 0, m
        assert viol == 0 and late == 0 and alive == 0 and depth == 0, m
        assert comms == (2 if two_comms else 1) * (1 if len(parts) > 1 else world), m
        assert sends > 0 and recvs > 0 and sbytes > 0 and rbytes > 0 and allred > 0, m
        if overlapped:
            assert p2p_streams >= 2, m          # migration records on the main stream, halos on the side stream
    if len(parts) > 1:
        # what all ranks sent is what all ranks received
        assert sum(int(q["mock"][4]) for q in parts) == sum(int(q["mock"][5]) for q in parts)
        assert sum(int(q["mock"][2]) for q in parts) == sum(int(q["mock"][3]) for q in parts)


@pytest.mark.parametrize("case,world,steps,fb,tol,calls,axis,exchange", [
    ("dam_break_3d_shipped", 2, 40, 8, 1e-9, 1, -1, "allreduce"),
    ("dam_break_3d_shipped", 2, 40, 4, 1e-5, 1, -1, "mailbox"),
    ("dam_break_3d_shipped", 3, 40, 8, 1e-9, 2, -1, "allreduce"),
    ("dam_break_3d_shipped", 4, 30, 8, 1e-9, 1, 0, "mailbox"),
    ("dam_break_3d_shipped", 4, 30, 4, 1e-5, 1, 0, "allreduce"),
    ("dam_break_2d", 2, 60, 8, 1e-9, 1, -1, "allreduce"),
    ("moving_square", 2, 150, 8, 1e-9, 1, 1, "allreduce"),          # particles cross the cut one by one: migration records of a few rows
    ("moving_square", 3, 120, 8, 1e-9, 1, 1, "mailbox"),
    ("dam_break_2d_mdbc", 2, 40, 8, 1e-9, 1, -1, "allreduce"),      # wide ghost layers, mDBC on ghost copies, serial pass 1
    ("dam_break_2d_mdbc", 3, 40, 8, 1e-9, 1, -1, "mailbox")])
def test_rccl_branch_rank_processes_match_one_device(mock_lib, case, world, steps, fb, tol, calls, axis, exchange, request, tmp_path):
    res = _spawn(mock_lib, world, lambda r: (case, steps, fb, str(tmp_path), calls, axis), extra_env={"SPHMI_EXCHANGE": exchange, "SPHMI_MBOX_TIMEOUT": "30"})
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
        assert "VIOLATION" not in e
    p, ref, progs = _reference(request, case, fb, steps, calls)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    _check_union(parts, p, ref, progs, fb, tol, 1)
    _check_mock(parts, world)
    for q in parts:
        assert int(q["info"][6]) == (1 if exchange == "mailbox" else 0)


ASYNC = {"MOCK_RCCL_ASYNC": "1", "MOCK_RCCL_ASYNC_DELAY_US": "800", "GPU_MAX_HW_QUEUES": "24", "MOCK_RCCL_TIMEOUT": "60"}


@pytest.mark.parametrize("case,world,steps,fb,tol,mode,exchange", [
    ("dam_break_3d_shipped", 2, 40, 8, 1e-9, "run", "allreduce"),
    ("dam_break_3d_shipped", 3, 40, 4, 1e-5, "run", "allreduce"),
    ("dam_break_3d_shipped", 4, 30, 8, 1e-9, "run", "mailbox"),
    ("moving_square", 2, 150, 8, 1e-9, "run", "allreduce"),
    ("dam_break_2d_mdbc", 3, 40, 8, 1e-9, "run", "allreduce"),
    ("dam_break_3d_shipped", 3, 40, 8, 1e-9, "run_slabs", "allreduce"),
    ("dam_break_2d_mdbc", 2, 40, 8, 1e-9, "run_slabs", "allreduce")])
def test_rccl_branch_under_rccl_timing(mock_lib, case, world, steps, fb, tol, mode, exchange, request, tmp_path):
    """$MOCK_RCCL_ASYNC=1: the double keeps RCCL's own timing — a call only queues; a proxy thread reads the send buffers when the stream reaches the
    call, writes the receive buffers, and releases the stream behind it.  A send buffer packed again too early, an unpack or an edge launch that does
    not wait for the stream the receive was posted on, a control that reads the maxima before the allreduce has landed: each would be wrong BYTES here
    (the default mode of the double completes every operation inside the call and cannot see them).  Same assertions as above."""
    extra = dict(ASYNC, SPHMI_EXCHANGE=exchange, SPHMI_MBOX_TIMEOUT="30")
    res = _spawn(mock_lib, world, lambda r: (case, steps, fb, str(tmp_path), 1, -1), mode=mode, extra_env=extra, timeout=300, n_procs=None if mode == "run" else 1)
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
        assert "VIOLATION" not in e
    p, ref, progs = _reference(request, case, fb, steps, 1)
    if mode == "run":
        parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
        _check_union(parts, p, ref, progs, fb, tol, 1)
    else:
        parts = [np.load(tmp_path / "rank0.npz")]
        _check_union(parts, p, ref, progs, fb, tol, world)
    _check_mock(parts, world, proxied=True)


def test_double_sees_stream_order_mistakes(mock_lib):
    """What the asynchronous mode is for, shown on two deliberate mistakes (tests/mock_rccl/stream_order_probe.py): a consumer stream that reads the
    receive buffer without waiting for the stream the receive was posted on, and a send buffer overwritten right after the call.  The default mode
    completes every operation inside the call and sees neither; under $MOCK_RCCL_ASYNC=1 both deliver wrong bytes, and the ordered caller stays right."""
    import json
    rounds = 8
    seen = {}
    for name, extra in (("default", {}), ("async", dict(ASYNC, MOCK_RCCL_ASYNC_DELAY_US="3000"))):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, os.path.join(MOCK_DIR, "stream_order_probe.py"), str(rounds)], env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stderr[-3000:]
        seen[name] = json.loads(r.stdout.strip().splitlines()[-1])
        assert seen[name]["violations"] == 0
    assert seen["default"]["by_proxy"] == 0 and seen["async"]["by_proxy"] == seen["async"]["groups"] == 3 * rounds
    assert seen["default"]["right"] == {"ordered": rounds, "early_read": rounds, "early_pack": rounds}, seen      # blind: every call is a full stop
    assert seen["async"]["right"]["ordered"] == rounds, seen
    if seen["async"]["right"]["early_read"] == rounds and seen["async"]["right"]["early_pack"] == rounds:
        pytest.skip("the runtime put the probe's two streams on one hardware queue: the racing stream waited behind the parked one")
    assert seen["async"]["right"]["early_read"] == 0 and seen["async"]["right"]["early_pack"] == 0, seen


def test_rccl_branch_with_one_communicator(mock_lib, request, tmp_path):
    """$SPHMI_RCCL_ONE_COMM=1: point-to-point traffic and the per-step allreduce on ONE communicator, called from two streams — legal only
    while every rank issues the calls in the same order, which is what the double's group-sequence check is about."""
    world, steps = 3, 40
    res = _spawn(mock_lib, world, lambda r: ("dam_break_3d_shipped", steps, 8, str(tmp_path), 1, -1), extra_env={"SPHMI_RCCL_ONE_COMM": "1"})
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, ref, progs = _reference(request, "dam_break_3d_shipped", 8, steps, 1)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    _check_union(parts, p, ref, progs, 8, 1e-9, 1)
    _check_mock(parts, world, two_comms=False)


@pytest.mark.parametrize("case,world,steps,fb,tol", [("dam_break_3d_shipped", 3, 40, 8, 1e-9), ("dam_break_2d_mdbc", 2, 40, 8, 1e-9), ("moving_square", 2, 100, 8, 1e-9)])
def test_rccl_branch_one_process_slabs(mock_lib, case, world, steps, fb, tol, request, tmp_path):
    """sphmi_create with a device list through ncclCommInitAll (twice: two communicators per slab): ONE host thread posts every slab's
    sends, receives and allreduces inside one group — the arrangement of a Julia process driving the GPUs of a node."""
    res = _spawn(mock_lib, world, lambda r: (case, steps, fb, str(tmp_path), 1, -1), mode="run_slabs", n_procs=1)
    (rc, o, e), = res
    assert rc == 0, e[-3000:]
    assert "VIOLATION" not in e
    p, ref, progs = _reference(request, case, fb, steps, 1)
    parts = [np.load(tmp_path / "rank0.npz")]
    _check_union(parts, p, ref, progs, fb, tol, world)
    _check_mock(parts, world)


def test_rccl_branch_forces_once(mock_lib, request, tmp_path):
    """sphmi_forces_once on rank-mode slabs: one more collective rebuild (migration, fresh ghost layers) and a forces-only pass; the union of
    the ranks' rows is the one-device hook's output, ID for ID."""
    from sphexample_amd.engine import make_engine
    world, steps, case = 3, 20, "dam_break_3d_shipped"
    res = _spawn(mock_lib, world, lambda r: (case, steps, 8, str(tmp_path), 1, -1), extra_env={"SPHMI_TEST_FORCES_ONCE": "1"})
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, s = request.getfixturevalue(case)
    ref = make_engine(p, s, device_float_bytes=8)
    ref.advance(1e9, max_steps=steps)
    drho, acc = ref.forces_once()
    ids = ref.download(("ID",))["ID"]
    o = np.argsort(ids, kind="stable")
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    gid = np.concatenate([q["ID_forces"] for q in parts])
    g = np.argsort(gid, kind="stable")
    np.testing.assert_array_equal(gid[g], ids[o])
    gd, ga = np.concatenate([q["drhodt"] for q in parts])[g], np.concatenate([q["acc"] for q in parts])[g]
    assert np.abs(gd - drho[o]).max() <= 1e-10 * np.abs(drho).max()
    assert np.abs(ga - acc[o]).max() <= 1e-10 * np.abs(acc).max()
    _check_mock(parts, world)


def test_rccl_branch_c3_flowing_four_ranks(mock_lib, tmp_path):
    """BASELINE config 3's size (1.06 M particles, flowing: rebuilds, migration and re-cuts inside the window) on four rank processes through
    the RCCL branch: halo messages of ≈1e5 records, capacities, the negotiation of counts — and the double's deadline instead of a hang."""
    from conftest import load_dam_break_3d_c3_flowing
    from sphexample_amd.engine import make_engine
    world, steps = 4, 50
    res = _spawn(mock_lib, world, lambda r: ("dam_break_3d_c3_flowing", steps, 4, str(tmp_path), 1, -1), timeout=900)
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, s = load_dam_break_3d_c3_flowing()
    ref = make_engine(p, s, device_float_bytes=4)
    pr = ref.advance(1e9, max_steps=steps)
    progs = np.array([[pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter, pr.total_time, pr.last_dt]], dtype=np.float64)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert pr.n_rebuilds >= 2
    _check_union(parts, p, ref, progs, 4, 1e-5, 1)
    _check_mock(parts, world)
    assert max(int(q["mock"][10]) for q in parts) > 1_000_000          # a halo message of more than a megabyte went through


def test_a_rank_that_stops_early_is_an_error_not_a_hang(mock_lib, tmp_path):
    """Rank 1 is told to take 20 steps, rank 0 forty: rank 0's per-step allreduce never meets its partner.  A real RCCL waits for ever;
    here sphmi_advance returns SPHMI_ERR_DEVICE and the text says which call of which rank was waiting."""
    res = _spawn(mock_lib, 2, lambda r: ("dam_break_2d", 40 if r == 0 else 20, 8, str(tmp_path), 1, -1), extra_env={"MOCK_RCCL_TIMEOUT": "5"}, timeout=300)
    # (rank 1 leaves its loop and enters the closing allreduce of sphmi_advance — one Int64 — while rank 0 posts the per-step one — four
    # UInt64: the double reports the mismatch, or, had rank 1 already gone, rank 0's deadline)
    text = res[0][2] + res[1][2]
    assert res[0][0] != 0 or res[1][0] != 0
    assert "[mock-rccl] VIOLATION" in text and ("entered ncclAllReduce with count" in text or "still waiting" in text), text[-3000:]
