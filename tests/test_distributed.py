"""Domain decomposition, CPU side: the Python planner the C++ planner is checked against (tests/test_multi_gpu.py), and the
launcher-side rendezvous of a one-process-per-GPU run over gloo with world_size 2 and 3.  The slab driver itself lives in
libsphmi.so and is tested through it: tests/test_multi_gpu.py (one handle, several slabs), tests/test_rank_mode.py (one
process per slab), tests/test_config_scale_gpu.py (at the sizes the metric is quoted on)."""
import os
import socket
import tempfile

import numpy as np
import pytest

from sphexample_amd._abi import make_config
from slab_planner_reference import SlabPlan, cell_x_of, choose_axis, step_control


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(fn, args=(world, _free_port(), d) + args, nprocs=world, join=True)
        return {f: open(os.path.join(d, f), "rb").read() for f in os.listdir(d)}


def test_slab_plan_balances_particles(dam_break_3d_shipped):
    p, s = dam_break_3d_shipped
    cx = cell_x_of(p.Position[:, 0], s.SimKernel.H_inv)
    for world in (2, 4):
        plan = SlabPlan.from_columns(cx, world)
        owner = plan.owner_of(cx)
        counts = np.bincount(owner, minlength=world)
        assert counts.sum() == len(p) and counts.min() > 0
        # the water column sits in the left quarter of the tank: a spatial cut would give rank 0 everything
        assert counts.max() < 2.2 * len(p) / world
        for r in range(world):
            sel = owner == r
            assert cx[sel].min() >= plan.cx_lo[r] and cx[sel].max() <= plan.cx_hi[r]
        assert all(plan.cx_hi[r] + 1 == plan.cx_lo[r + 1] for r in range(world - 1))


def test_choose_axis_prefers_the_evenly_filled_direction(dam_break_3d_shipped):
    """The 3-D dam break fills the whole width (y) of the tank but a short stretch of its length (x): slabs along
    y balance better than slabs along x; z (few layers) cannot host 8 ranks at this resolution."""
    p, s = dam_break_3d_shipped
    cols = [cell_x_of(p.Position[:, a], s.SimKernel.H_inv) for a in range(3)]
    loads = {}
    for ax in range(3):
        plan = SlabPlan.from_columns(cols[ax], 2)
        loads[ax] = np.bincount(plan.owner_of(cols[ax]), minlength=2).max()
    ax = choose_axis(cols, 2)
    assert loads[ax] <= 1.01 * min(loads.values())               # within 1 % the thinner ghost layer decides
    with pytest.raises(ValueError):
        choose_axis([np.zeros(10, dtype=np.int64)], 2)          # one column cannot be split


def test_cuts_balance_work_not_counts():
    """particle_work = candidates in the 3^D cells around a particle (what it costs in the neighbour kernel); best_cuts
    = the exact lightest-heaviest-slab partition.  On the generated dam break, count-balanced cuts along x leave the
    rank with the dry walls ≈10 % short of work; the work-balanced plan picks y and stays within a column's worth."""
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from slab_planner_reference import best_cuts, particle_work
    dp = 0.0085
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    cols = [cell_x_of(p.Position[:, a], s.SimKernel.H_inv) for a in range(3)]
    w = particle_work(cols)
    # brute-force check of the measure on a few particles
    lin = {}
    key = list(zip(*cols))
    from collections import Counter
    cnt = Counter(key)
    for i in (0, len(p) // 3, len(p) - 1):
        c = key[i]
        assert w[i] == sum(cnt.get((c[0] + a, c[1] + b, c[2] + d), 0) for a in (-1, 0, 1) for b in (-1, 0, 1) for d in (-1, 0, 1))
    assert w[p.Type == 1].mean() > 3 * w[p.Type != 1].mean()
    ax = choose_axis(cols, 2, 2, w)
    plan = SlabPlan.from_columns(cols[ax], 2, 2, w)
    load = np.bincount(plan.owner_of(cols[ax]), weights=w, minlength=2)
    by_count = SlabPlan.from_columns(cols[0], 2)
    load0 = np.bincount(by_count.owner_of(cols[0]), weights=w, minlength=2)
    # 23 columns along y at this resolution: one column is 4 % of the work
    assert ax == 1 and load.max() / load.mean() < 1.05 < load0.max() / load0.mean()
    # best_cuts against exhaustive search on a small histogram, with and without bounds
    rng = np.random.default_rng(3)
    h = rng.integers(0, 50, size=12).astype(float)
    import itertools
    def brute(world, width, lo=None, hi=None):
        best = None
        for cs in itertools.combinations(range(1, 12), world - 1):
            e = (0,) + cs + (12,)
            if any(e[k + 1] - e[k] < width for k in range(world)):
                continue
            if lo and any(not (lo[r] <= cs[r - 1] <= hi[r]) for r in range(1, world)):
                continue
            m = max(h[e[k]:e[k + 1]].sum() for k in range(world))
            best = m if best is None else min(best, m)
        return best
    for world, width in ((2, 2), (3, 2), (4, 3), (4, 2)):
        cs = best_cuts(h, world, width)
        e = [0] + cs + [12]
        assert max(h[e[k]:e[k + 1]].sum() for k in range(world)) == brute(world, width)
    lo, hi = [0, 3, 6, 12], [0, 5, 9, 12]
    cs = best_cuts(h, 3, 2, lo, hi)
    e = [0] + cs + [12]
    assert lo[1] <= cs[0] <= hi[1] and lo[2] <= cs[1] <= hi[2]
    assert max(h[e[k]:e[k + 1]].sum() for k in range(3)) == brute(3, 2, lo, hi)
    assert best_cuts(h, 7, 2) is None


def test_slab_plan_keeps_halo_wide_slabs():
    """With mDBC the ghost layers are several columns wide and come from ONE neighbour: slabs stay at least that wide,
    in the initial cuts and in every re-cut."""
    from slab_planner_reference import SlabPlan, choose_axis
    cx = np.repeat(np.arange(20), 10)
    plan = SlabPlan.from_columns(cx, 4, min_width=5)
    assert plan.cuts() == [5, 10, 15] and plan.min_width == 5
    with pytest.raises(ValueError):
        SlabPlan.from_columns(cx, 5, min_width=5)
    hist = np.zeros(20, dtype=np.int64); hist[:6] = 100                     # everything piled up on the left
    new = plan.recut(0, hist)
    w = np.diff([0] + new.cuts() + [20])
    assert (w >= 5).all() and new.min_width == 5
    # an axis too short for the halo is not chosen even if it balances better
    cols = [np.repeat(np.arange(8), 25), np.tile(np.arange(25), 8)]
    assert choose_axis(cols, 2, [5, 2]) == 1 and choose_axis(cols, 2, [2, 2]) in (0, 1)
    with pytest.raises(ValueError):
        choose_axis(cols, 2, [5, 13])


def test_recut_keeps_migration_between_neighbours():
    """SlabPlan.recut: equal-count cuts for the current histogram, every cut between its old neighbours, slabs ≥ 2
    columns; repeated re-cuts converge to the balanced plan."""
    rng = np.random.default_rng(0)
    cx = rng.integers(0, 40, 100000)
    plan = SlabPlan.from_columns(cx, 4)
    cx2 = np.concatenate([cx, rng.integers(0, 10, 60000)])          # the fluid piles up on the left
    hist = np.bincount(cx2, minlength=40)
    before = np.bincount(plan.owner_of(cx2), minlength=4).max()
    for _ in range(3):
        new = plan.recut(0, hist)
        old_c, new_c = [0] + plan.cuts() + [40], [0] + new.cuts() + [40]
        assert all(old_c[r - 1] <= new_c[r] <= old_c[r + 1] for r in range(1, 4))
        assert all(new_c[r + 1] - new_c[r] >= 2 for r in range(4))
        # a particle changes rank by at most one
        assert np.abs(new.owner_of(cx2) - plan.owner_of(cx2)).max() <= 1
        plan = new
    assert np.bincount(plan.owner_of(cx2), minlength=4).max() < 0.6 * before
    assert plan.recut(0, hist).cuts() == plan.cuts()                   # converged


def test_step_control_matches_oracle_dt(dam_break_2d):
    """The host-side Δt / Δx arithmetic used by the distributed driver = the oracle's (TimeStepping.jl:30-43)."""
    from oracle.oracle import make_oracle
    from conftest import perturbed
    p, s = dam_break_2d
    q = perturbed(p, seed=2)
    q.Acceleration[:] = np.random.default_rng(0).normal(size=q.Acceleration.shape)
    o = make_oracle(q, s)
    cfg = make_config(len(q), s.SimConstants, s.SimKernel, s.SimMetaData, s.SimViscosity, s.SimDensityDiffusion)
    k = s.SimKernel
    visc = np.abs(k.h * (q.Velocity * q.Position).sum(1) / ((q.Position ** 2).sum(1) + k.eta2)).max()
    red = np.array([0.0, visc, (q.Acceleration ** 2).sum(1).max(), 0.0])
    dx, dt, rebuild = step_control(red, 0.0, cfg)
    assert dt == pytest.approx(o.delta_t(), rel=1e-13)
    assert dx == 0.0 and not rebuild
    dx, _, rebuild = step_control(np.array([(k.h / 3.9) ** 2, visc, 1.0, 0.0]), 0.0, cfg)
    assert rebuild and dx >= k.h


def _rendezvous_worker(rank, world, port, out_dir):
    """CPU-only: the launcher-side glue bench.py runs under torchrun (sphexample_amd/rendezvous.py) over gloo."""
    from sphexample_amd.rendezvous import Rendezvous, create_rank_engine
    os.environ.pop("MASTER_PORT", None)
    rdv = Rendezvous(rank, world, master_addr="127.0.0.1", master_port=port)
    ok = rdv.broadcast_bytes(b"\x07" * 128 if rank == 0 else None) == b"\x07" * 128
    ok &= rdv.max(1.0 + rank) == float(world)
    ok &= rdv.all_ok(True) and not rdv.all_ok(rank != world - 1)
    ok &= rdv.gather_strings(f"r{rank}") == [f"r{k}" for k in range(world)]
    rdv.barrier()
    # the collective create: one rank failing is learnt by every rank, with its text; nobody hangs, nobody keeps an engine
    os.environ["SPHMI_TRANSPORT"] = "shm"                       # (no RCCL id needed: any 128 bytes)

    class Fake:
        closed = False
        def close(self): self.closed = True

    made = []
    def make(uid):
        assert len(uid) == 128
        if rank == world - 1:
            raise RuntimeError("no device for this rank")
        made.append(Fake()); return made[-1]
    # (Fake engines: the default local check — libsphmi.so loads, RCCL binds — is not theirs; a caller with its own factory switches it off)
    eng, errs = create_rank_engine(rdv, make, local_check=None)
    ok &= eng is None and errs == [f"rank {world - 1}: no device for this rank"] and all(f.closed for f in made)
    eng, errs = create_rank_engine(rdv, lambda uid: Fake(), local_check=None)
    ok &= isinstance(eng, Fake) and errs == []
    # a caller's own id: asked for ONCE, on rank 0 (ncclGetUniqueId starts a bootstrap root per call)
    ids = []
    eng, errs = create_rank_engine(rdv, lambda uid: Fake(), local_check=None, make_id=lambda: ids.append(1) or b"\x05" * 128)
    ok &= isinstance(eng, Fake) and len(ids) == (1 if rank == 0 else 0)
    # the default local check is injectable and its failure is a LOCAL failure
    def broken(shm):
        raise RuntimeError("libsphmi.so is missing")
    eng, errs = create_rank_engine(rdv, lambda uid: Fake(), local_check=broken if rank == 1 else None)
    ok &= eng is None and errs == ["rank 1 (local set-up): libsphmi.so is missing"]
    # a rank that fails in its LOCAL phase (no device, no memory, library missing) is learnt by every rank BEFORE anyone enters the
    # collective set-up: make() — where a real rank would sit in ncclCommInitRank waiting for the missing peer — is never called
    entered = []
    def preflight():
        if rank == 0:
            raise RuntimeError("no memory for this slab")
    eng, errs = create_rank_engine(rdv, lambda uid: entered.append(1) or Fake(), preflight=preflight, local_check=None)
    ok &= eng is None and errs == ["rank 0 (local set-up): no memory for this slab"] and not entered
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    rdv.close()


@pytest.mark.parametrize("world", [2, 3])
def test_rendezvous_over_gloo(world):
    out = _spawn(_rendezvous_worker, world)
    assert out == {f"ok{r}": b"1" for r in range(world)}


def test_rendezvous_needs_a_port_for_more_than_one_rank(monkeypatch):
    from sphexample_amd.rendezvous import Rendezvous, free_port
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        Rendezvous(0, 2)
    assert free_port() != free_port() or True                 # (two free ports may coincide once released; the call works)
