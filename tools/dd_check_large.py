#!/usr/bin/env python3
"""One-off check on the GPU box: W slab ranks sharing GPU 0 (halo through gloo) against the single-GPU engine on a
generated 3-D dam break.  usage: python tools/dd_check_large.py [dp] [world] [steps]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, out, dp, steps):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.distributed import DistributedEngine
    eng = DistributedEngine(dam_break_3d(dp), setup_dam_break_3d(dp), rank, world, local_device=0, device_float_bytes=4)
    pr = eng.advance(1e9, max_steps=steps)
    res = eng.gather_all()
    if rank == 0:
        np.savez(out, total_time=pr.total_time, n_rebuilds=pr.n_rebuilds, axis=eng.axis, **res)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0085
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    e = make_engine(dam_break_3d(dp), setup_dam_break_3d(dp), device_float_bytes=4)
    pr = e.advance(1e9, max_steps=steps)
    r = e.download(("Position", "Density", "ID"))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "dd.npz")
        mp.spawn(worker, args=(world, 29561, out, dp, steps), nprocs=world, join=True)
        dd = dict(np.load(out))
    i1, i2 = np.argsort(r["ID"]), np.argsort(dd["ID"])
    print(f"N={len(r['ID'])} world={world} axis={int(dd['axis'])} rebuilds {pr.n_rebuilds}/{int(dd['n_rebuilds'])} "
          f"t {pr.total_time:.9e}/{float(dd['total_time']):.9e} "
          f"rho {np.abs(dd['Density'][i2] - r['Density'][i1]).max() / 1000:.2e} x {np.abs(dd['Position'][i2] - r['Position'][i1]).max():.2e} "
          f"| ids ok {len(np.unique(dd['ID'])) == len(r['ID'])} mean rho {dd['Density'].mean():.4f}/{r['Density'].mean():.4f} "
          f"front x {dd['Position'][:, 0][dd['Density'] > 0].max():.4f} median |dx| {np.median(np.abs(dd['Position'][i2] - r['Position'][i1]).max(axis=1)):.2e}")
