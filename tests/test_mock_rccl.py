"""The RCCL test double checks what it claims to check (CPU processes, host-buffer mode; no GPU).

tests/mock_rccl/mock_rccl.cpp stands in for librccl.so behind $SPHMI_RCCL_LIB so that the RCCL branch of the slab driver
(csrc/sphmi_multi.h) runs with several ranks on ONE GPU (tests/test_mock_rccl_gpu.py).  A double that accepts everything would
prove nothing: here every rule it enforces is broken once, on purpose, and must come back as an error with a text that names
the ranks and the message — not as a hang."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MOCK_DIR = os.path.join(HERE, "mock_rccl")
WORKER = os.path.join(MOCK_DIR, "worker.py")


@pytest.fixture(scope="module")
def mock_lib():
    sys.path.insert(0, MOCK_DIR)
    try:
        import build as mock_build
        return mock_build.build()
    finally:
        sys.path.remove(MOCK_DIR)


def _spawn(scen, world, timeout_s="3", launch=None):
    uid = os.urandom(128).hex()
    env = dict(os.environ, MOCK_RCCL_HOST_BUFFERS="1", MOCK_RCCL_TIMEOUT=timeout_s)
    ranks = range(world) if launch is None else launch
    procs = [subprocess.Popen([sys.executable, WORKER, scen, uid, str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in ranks]
    out = []
    try:
        for pr in procs:
            o, e = pr.communicate(timeout=120)
            out.append((pr.returncode, o, e))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return out


@pytest.mark.parametrize("world", [2, 3, 4])
def test_matched_traffic_passes(mock_lib, world):
    """Neighbour messages longer than a channel's ring in both directions inside one group, MAX / SUM allreduces, a zero-byte
    message; the counters add up and no violation is recorded."""
    for rc, o, e in _spawn("ok", world, timeout_s="30"):
        assert rc == 0, (o, e[-2000:])
        assert "VIOLATION" not in e


def test_exports_the_symbols_libsphmi_binds(mock_lib):
    """Exactly the names csrc/sphmi_multi.h resolves with dlsym (Rccl::get): a renamed entry point would make the substitute unloadable."""
    import re
    src = open(os.path.join(os.path.dirname(HERE), "sphexample_amd", "csrc", "sphmi_multi.h")).read()
    wanted = set(re.findall(r'sym\("(nccl\w+)"', src))
    assert len(wanted) == 11
    out = subprocess.run(["nm", "-D", "--defined-only", mock_lib], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert wanted <= have, wanted - have


@pytest.mark.parametrize("scen,needle", [
    ("size_mismatch", "posted 100 bytes"),
    ("group_mismatch", "call sequences differ"),
    ("coll_mismatch", "entered ncclAllReduce with count"),
    ("interleave_mismatch", "call sequences differ")])
def test_mismatched_ranks_fail_with_a_text(mock_lib, scen, needle):
    """Byte counts that differ, groups cut differently, collectives with different counts, a collective on one side of a message
    on rank 0 and on the other side on rank 1: both ranks come back with an error (the one that saw it: the text; its peer: the
    same text through the poisoned segment, or its own deadline)."""
    res = _spawn(scen, 2)
    assert all(rc == 0 for rc, _, _ in res), res
    assert any(needle in o + e for _, o, e in res), res


def test_a_send_nobody_receives_runs_into_the_deadline(mock_lib):
    res = _spawn("missing_recv", 2, timeout_s="2")
    rc, o, e = res[0]
    assert rc == 0 and "still waiting" in o + e and "ncclSend of rank 0 to rank 1 (100 bytes" in o + e, res


def test_a_missing_peer_fails_the_communicator(mock_lib):
    (rc, o, e), = _spawn("missing_peer", 2, timeout_s="2", launch=[0])
    assert rc == 0 and "waited 2 s for its peers (1 of 2 arrived)" in o + e


@pytest.mark.parametrize("scen,needle", [("open_group_at_destroy", "inside an open ncclGroupStart"), ("group_end_without_start", "without ncclGroupStart")])
def test_unbalanced_groups_are_violations(mock_lib, scen, needle):
    (rc, o, e), = _spawn(scen, 1)
    assert rc == 0 and needle in o + e, (o, e)
