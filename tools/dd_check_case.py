#!/usr/bin/env python3
"""GPU box: `world` slab ranks sharing GPU 0 (halo through gloo) against the single engine on one of the test cases.
usage: python tools/dd_check_case.py case steps [fb] [world] [axis]      (case: a tests/conftest.py load_* name)"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, out, case, steps, fb, axis):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest
    from sphexample_amd.distributed import DistributedEngine
    p, s = getattr(conftest, "load_" + case)()
    eng = DistributedEngine(p, s, rank, world, local_device=0, device_float_bytes=fb, axis=axis)
    if hasattr(p, "geometries") and not os.environ.get("NO_MOTION"):
        eng.set_motions(p.geometries)
    pr = eng.advance(1e9, max_steps=steps)
    own = eng.download_owned()
    res = eng.gather_all()
    counts = [None] * world
    dist.all_gather_object(counts, len(own["ID"]))
    if rank == 0:
        np.savez(out, total_time=pr.total_time, n_rebuilds=pr.n_rebuilds, axis=eng.axis, halo_width=eng.halo_width,
                 n_recuts=eng.n_recuts, counts=np.array(counts), **res)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import conftest
    from sphexample_amd.engine import make_engine
    case, steps = sys.argv[1], int(sys.argv[2])
    fb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    world = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    axis = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "-" else None
    p, s = getattr(conftest, "load_" + case)()
    e = make_engine(p, s, device_float_bytes=fb)
    if hasattr(p, "geometries") and not os.environ.get("NO_MOTION"):
        e.set_motions(p.geometries)
    pr = e.advance(1e9, max_steps=steps)
    r = e.download(("Position", "Density", "ID"))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "dd.npz")
        mp.spawn(worker, args=(world, 29563, out, case, steps, fb, axis), nprocs=world, join=True)
        dd = dict(np.load(out))
    i1, i2 = np.argsort(r["ID"]), np.argsort(dd["ID"])
    print(f"{case} N={len(r['ID'])} fb={fb} world={world} axis={int(dd['axis'])} halo={int(dd['halo_width'])} owned={dd['counts'].tolist()} "
          f"recuts={int(dd['n_recuts'])} rebuilds {pr.n_rebuilds}/{int(dd['n_rebuilds'])} t {pr.total_time:.9e}/{float(dd['total_time']):.9e} "
          f"rho {np.abs(dd['Density'][i2] - r['Density'][i1]).max() / 1000:.2e} x {np.abs(dd['Position'][i2] - r['Position'][i1]).max():.2e}", flush=True)
