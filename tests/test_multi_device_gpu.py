"""More than one GPU: these tests ACTIVATE THEMSELVES on a box whose torch.cuda.device_count() is ≥ 2 and skip on the one-GPU
boxes every round so far has had ("nothing in the suite will exercise RCCL with more than one rank
even when a multi-GPU box appears").  What they hold the decomposition of src/SPHCellList.jl:727-805 to:

  * ONE handle over a device list — devices = [0, 1] and [0 … n−1] — against the one-device handle: ID for ID the same sorted
    order and cells, the same Δt / rebuild sequence and IndexCounter, ρ and x within 1e-9 (fp64) / 1e-5 (fp32); transport = RCCL
    (ncclCommInitAll, ncclSend / ncclRecv between slab neighbours, the 4-word ncclAllReduce per step) and, with
    SPHMI_TRANSPORT=local, peer copies over xGMI;
  * sphmi_create_rank over REAL RCCL — 2, 4 and 8 processes, one GPU each, at BASELINE config 3's size — against the one-device
    result: every particle owned exactly once, every rank reports the loop counters of the whole run;
  * `python bench.py --gpus N` (no launcher) and `--single-process` produce their lines with the RCCL transport named.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import flowing

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "rank_worker.py")


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:                                   # noqa: BLE001
        return 0


N_GPUS = _n_gpus()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_GPUS < 2, reason=f"needs at least two GPUs (this box has {N_GPUS})")]
WORLDS = [w for w in (2, 4, 8) if w <= N_GPUS]


def _by_id(st):
    o = np.argsort(st["ID"], kind="stable")
    return {k: v[o] for k, v in st.items()}


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _one_handle_vs_one_device(p, s, fb, tol, devices, calls, transport=None, monkeypatch=None, exchange=None):
    from sphexample_amd.engine import make_engine
    if transport:
        monkeypatch.setenv("SPHMI_TRANSPORT", transport)
    if exchange:
        monkeypatch.setenv("SPHMI_EXCHANGE", exchange); monkeypatch.setenv("SPHMI_MBOX_TIMEOUT", "30")
    ref = make_engine(p, s, device_float_bytes=fb)
    dd = make_engine(p, s, device_float_bytes=fb, devices=devices)
    info = dd.multi_info()
    assert info.world == len(devices) and info.n_local == len(devices)
    assert info.transport == (0 if transport == "local" else 1)          # 1 = RCCL, 0 = stream-ordered (peer) copies
    assert info.reserved == (1 if exchange == "mailbox" else 0)          # … and the per-step maxima through the collective / the mailboxes
    for steps in calls:
        pr, pd = ref.advance(1e9, max_steps=steps), dd.advance(1e9, max_steps=steps)
        assert (pd.iteration, pd.steps_done, pd.n_rebuilds, pd.index_counter) == (pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter)
        assert pd.total_time == pytest.approx(pr.total_time, rel=1e-12 if fb == 8 else 1e-6)
        assert pd.last_dt == pytest.approx(pr.last_dt, rel=1e-12 if fb == 8 else 1e-5)
    fields = ("Position", "Density", "ID", "Velocity", "Cells")
    r, d = ref.download(fields), dd.download(fields)
    assert dd.owned_count() == len(p)
    np.testing.assert_array_equal(d["ID"], r["ID"])
    np.testing.assert_array_equal(d["Cells"], r["Cells"])
    np.testing.assert_array_equal(dd.unique_cells(), ref.unique_cells())
    assert _rel(d["Density"], r["Density"]) < tol and _rel(d["Position"], r["Position"]) < tol
    ref.close(); dd.close()
    return pr


@pytest.mark.parametrize("case,steps,fb,tol", [("dam_break_3d_shipped", 40, 8, 1e-9), ("dam_break_3d_shipped", 40, 4, 1e-5),
                                               ("moving_square", 150, 8, 1e-9), ("dam_break_2d_mdbc", 40, 8, 1e-9), ("duckling", 12, 8, 1e-9)])
@pytest.mark.parametrize("transport", [None, "local"])
def test_two_devices_in_one_handle_match_one_device(case, steps, fb, tol, transport, request, monkeypatch):
    p, s = request.getfixturevalue(case)
    _one_handle_vs_one_device(p, s, fb, tol, [0, 1], (steps // 2, steps - steps // 2), transport, monkeypatch)


@pytest.mark.parametrize("case,steps,fb,tol", [("dam_break_3d_shipped", 60, 8, 1e-9), ("moving_square", 150, 8, 1e-9)])
def test_two_devices_in_one_handle_with_the_mailbox_exchange(case, steps, fb, tol, request, monkeypatch):
    """$SPHMI_EXCHANGE=mailbox between two DEVICES of one process (peer access, the maxima posted over xGMI) instead of ncclAllReduce:
    the first run of this path on real peers — on one GPU it is covered by the rank processes of tests/test_rank_mode.py."""
    p, s = request.getfixturevalue(case)
    _one_handle_vs_one_device(p, s, fb, tol, [0, 1], (steps // 2, steps - steps // 2), None, monkeypatch, exchange="mailbox")


@pytest.mark.parametrize("fb,tol", [(8, 1e-9), (4, 1e-5)])
def test_every_device_in_one_handle_matches_one_device(fb, tol, dam_break_3d_shipped):
    p, s = dam_break_3d_shipped
    _one_handle_vs_one_device(flowing(p, seed=5, shear=2.0, base=1.0), s, fb, tol, list(range(min(N_GPUS, 8))), (20, 40))


def test_c3_every_device_in_one_handle():
    """BASELINE config 3's lattice in the streaming state (a rebuild every ≈33 steps) on all GPUs of the box: collective rebuilds,
    migration across the cuts and the halo of every pass over RCCL."""
    from conftest import load_dam_break_3d_c3_flowing
    p, s = load_dam_break_3d_c3_flowing()
    pr = _one_handle_vs_one_device(p, s, 4, 1e-5, list(range(min(N_GPUS, 8))), (20, 60))
    assert pr.n_rebuilds >= 3


def _spawn_rccl(world, args_of, timeout=900):
    from sphexample_amd.engine import rccl_unique_id
    uid = rccl_unique_id().hex()
    env = {k: v for k, v in os.environ.items() if k != "SPHMI_TRANSPORT"}
    env.update(SPHMI_TEST_DEVICE_PER_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
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


@pytest.mark.parametrize("exchange", ["allreduce", "mailbox"])
@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case,steps,fb,tol", [("dam_break_3d_shipped", 40, 8, 1e-9), ("dam_break_3d_c3_flowing", 50, 4, 1e-5)])
def test_rank_mode_over_rccl_matches_one_device(world, case, steps, fb, tol, exchange, tmp_path, monkeypatch):
    """sphmi_create_rank, one process per GPU, the peers behind RCCL (info.transport == 1): the launch shape of
    `torchrun bench.py --gpus N`."""
    import conftest
    from sphexample_amd.engine import make_engine
    if case == "dam_break_3d_shipped" and world > 4:
        pytest.skip("17 k particles: too few cell columns for eight slabs")
    monkeypatch.setenv("SPHMI_EXCHANGE", exchange); monkeypatch.setenv("SPHMI_MBOX_TIMEOUT", "60")      # (the workers inherit it: mailboxes through hipIpc)
    res = _spawn_rccl(world, lambda r: ("run", case, steps, fb, str(tmp_path), 1, -1))
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
    p, s = getattr(conftest, "load_" + case)()
    ref = make_engine(p, s, device_float_bytes=fb)
    pr = ref.advance(1e9, max_steps=steps)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for q in parts:
        np.testing.assert_array_equal(q["prog"][0, :4], [pr.iteration, pr.steps_done, pr.n_rebuilds, pr.index_counter])
        np.testing.assert_allclose(q["prog"][0, 4:], [pr.total_time, pr.last_dt], rtol=1e-12 if fb == 8 else 1e-5)
        assert tuple(q["info"][:3]) == (world, 1, 1)                    # one local slab, RCCL
        assert int(q["info"][6]) == (1 if exchange == "mailbox" else 0)
    ids = np.concatenate([q["ID"] for q in parts])
    assert len(ids) == len(p) and len(np.unique(ids)) == len(p)        # every particle owned exactly once
    got = _by_id({k: np.concatenate([q[k] for q in parts]) for k in ("ID", "Density", "Position", "Velocity")})
    r = _by_id(ref.download(("ID", "Density", "Position", "Velocity")))
    assert _rel(got["Density"], r["Density"]) < tol and _rel(got["Position"], r["Position"]) < tol


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", WORLDS[:2])
@pytest.mark.parametrize("how", ["no-launcher", "single-process", "torchrun"])
def test_bench_lines_on_several_gpus(world, how):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SPHMI_TRANSPORT")}
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--dp", "0.012", "--precondition-ms", "0"]
    if how == "torchrun":
        from sphexample_amd.rendezvous import free_port
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port())] + tail
    else:
        cmd = [sys.executable] + tail + (["--single-process"] if how == "single-process" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == world and j["value"] > 0
    assert "RCCL" in j["config"]["parallelism"] and "FALLBACK" not in j["config"]["parallelism"] and "SHARED-MEMORY" not in j["config"]["parallelism"]
    assert ("self-spawned" in j["config"]["launch"]) == (how == "no-launcher")
