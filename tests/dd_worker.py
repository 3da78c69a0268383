"""Worker for the 2-rank domain-decomposition tests (spawned by torch.multiprocessing)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def comm_worker(rank, world, port, out_dir):
    """CPU-only: the point-to-point / allreduce plumbing of _Comm over gloo."""
    import torch
    dist = _init(rank, world, port)
    from sphexample_amd.distributed import _Comm
    comm = _Comm(rank, world, torch.device("cpu"))
    ok = True
    g = comm.allreduce_max(np.array([rank + 1.0, 5.0 - rank, 0.25, 0.0]))
    ok &= np.allclose(g, [world, 5.0, 0.25, 0.0])
    # counts: every rank tells its neighbours how much it will send
    nl, nr = comm.exchange_counts(10 + rank, 20 + rank)
    ok &= nl == (0 if rank == 0 else 20 + rank - 1)
    ok &= nr == (0 if rank == world - 1 else 10 + rank + 1)
    # payloads of different sizes in both directions
    mk = lambda n, v: torch.full((n,), v, dtype=torch.uint8)  # noqa: E731
    sl = mk(10 + rank, 100 + rank) if rank > 0 else None
    sr = mk(20 + rank, 200 + rank) if rank < world - 1 else None
    rl, rr = comm.exchange(sl, sr, nl, nr)
    if rank > 0:
        ok &= rl is not None and rl.numel() == nl and bool((rl == 200 + rank - 1).all())
    if rank < world - 1:
        ok &= rr is not None and rr.numel() == nr and bool((rr == 100 + rank + 1).all())
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def engine_worker(rank, world, port, out_dir, case, steps, fb, axis=None, overlap=True, recut=1.05, cut_shift=0, calls=1):
    """2 ranks sharing GPU 0 (gloo staging): slab engines vs nothing — rank 0 stores the gathered result."""
    dist = _init(rank, world, port)
    import conftest
    from sphexample_amd.distributed import DistributedEngine
    p, s = getattr(conftest, "load_" + case)()
    plan = None
    if cut_shift:
        # start from deliberately unbalanced cuts (the balanced ones moved by `cut_shift` columns)
        from sphexample_amd.distributed import SlabPlan, cell_x_of
        cx = cell_x_of(p.Position[:, axis], s.SimKernel.H_inv)
        b = SlabPlan.from_columns(cx, world)
        INF = 1 << 30
        c = [x + cut_shift for x in b.cuts()]
        plan = SlabPlan([-INF] + c, [x - 1 for x in c] + [INF])
    eng = DistributedEngine(p, s, rank, world, local_device=0, plan=plan, device_float_bytes=fb, axis=axis, overlap=overlap, recut_imbalance=recut)
    if hasattr(p, "geometries"):
        eng.set_motions(p.geometries)
    for _ in range(calls):            # every call re-arms Δx = 1 + h: a rebuild (and a chance to re-cut) at its first step
        pr = eng.advance(1e9, max_steps=steps // calls)
    res = eng.gather_all()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dd.npz"), iteration=pr.iteration, total_time=pr.total_time,
                 n_rebuilds=pr.n_rebuilds, axis=eng.axis, n_recuts=eng.n_recuts, halo_width=eng.halo_width, **res)
    dist.barrier()
    dist.destroy_process_group()


def cost_worker(rank, world, port, out_dir, case, axis):
    """Per-column work of sphmi_dd_column_cost (device, cell list of the first rebuild) summed over the ranks."""
    dist = _init(rank, world, port)
    import ctypes as C
    import torch
    import conftest
    from sphexample_amd.distributed import DistributedEngine
    p, s = getattr(conftest, "load_" + case)()
    eng = DistributedEngine(p, s, rank, world, local_device=0, device_float_bytes=8, axis=axis)
    eng.advance(1e9, max_steps=1)                 # the first control asks for the rebuild: cell list of the INITIAL positions
    col0, ncols = -8, 400
    with torch.cuda.stream(eng._main):
        cost = torch.zeros(ncols, dtype=torch.int64, device=eng.device)
        eng._call("dd_column_cost", C.c_int64(col0), C.c_int32(ncols), C.c_void_p(cost.data_ptr()))
        eng._main.synchronize()
    parts = [None] * world if rank == 0 else None
    dist.gather_object(cost.cpu().numpy(), parts, dst=0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "cost.npz"), cost=np.sum(parts, axis=0), col0=col0, axis=eng.axis)
    dist.barrier()
    dist.destroy_process_group()
