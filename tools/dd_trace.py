#!/usr/bin/env python3
"""GPU box: where and when two slab ranks (sharing GPU 0, halo through gloo) leave the single engine.
usage: python tools/dd_trace.py case axis step,step,…      (case: a tests/conftest.py load_* name)
env: JITTER=<pos scale> (fluid jitter: no lattice ties), NO_MOTION=1, NOSHIFT=1, ARTVISC=1, KCUT=<k>, DUMP=<n worst particles>
Found the in-cell order of migrants (order tags, sphmi_rebuild.h) on example/MovingSquare2d.jl."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CASE, AXIS = sys.argv[1], int(sys.argv[2])

def tweak(s):
    import dataclasses
    from sphexample_amd.config import ArtificialViscosity, NoShifting
    if os.environ.get("NOSHIFT"): s = dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, SMode=NoShifting))
    if os.environ.get("KCUT"):
        from sphexample_amd import SPHKernelInstance, WendlandC2
        s = dataclasses.replace(s, SimKernel=SPHKernelInstance(2, WendlandC2(), dx=s.SimConstants.dx, k=float(os.environ["KCUT"])))
    if os.environ.get("ARTVISC"): s = dataclasses.replace(s, SimViscosity=ArtificialViscosity())
    return s
MARKS = [int(x) for x in sys.argv[3].split(",")]

def worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import conftest
    from sphexample_amd.distributed import DistributedEngine
    p, s = getattr(conftest, "load_" + CASE)()
    s = tweak(s)
    g = getattr(p, "geometries", None) if not os.environ.get("NO_MOTION") else None
    if os.environ.get("JITTER"): p = conftest.perturbed(p, seed=1, vel_scale=0.0, rho_scale=0.0, pos_scale=float(os.environ["JITTER"])); p.geometries = g
    eng = DistributedEngine(p, s, rank, world, local_device=0, device_float_bytes=8, axis=AXIS)
    if g is not None: eng.set_motions(g)
    done = 0
    for m in MARKS:
        pr = eng.advance(1e9, max_steps=m - done); done = m
        own = eng.download_owned()
        cnt = [None] * world
        dist.all_gather_object(cnt, len(own["ID"]))
        res = eng.gather_all()
        if rank == 0:
            np.savez(os.path.join(out, f"dd{m}.npz"), cuts=np.array(eng.plan.cuts()), t=pr.total_time, cnt=np.array(cnt), **res)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import conftest
    from sphexample_amd.engine import make_engine
    from sphexample_amd.distributed import cell_x_of
    p, s = getattr(conftest, "load_" + CASE)()
    s = tweak(s)
    g = getattr(p, "geometries", None) if not os.environ.get("NO_MOTION") else None
    if os.environ.get("JITTER"): p = conftest.perturbed(p, seed=1, vel_scale=0.0, rho_scale=0.0, pos_scale=float(os.environ["JITTER"])); p.geometries = g
    e = make_engine(p, s, device_float_bytes=8)
    if g is not None: e.set_motions(g)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(worker, args=(2, 29571, d), nprocs=2, join=True)
        done = 0
        for m in MARKS:
            pr = e.advance(1e9, max_steps=m - done); done = m
            r = e.download(("Position", "Density", "ID", "Type", "Velocity"))
            dd = dict(np.load(os.path.join(d, f"dd{m}.npz")))
            i1, i2 = np.argsort(r["ID"]), np.argsort(dd["ID"])
            dr = np.abs(dd["Density"][i2] - r["Density"][i1]); dx = np.abs(dd["Position"][i2] - r["Position"][i1]).max(axis=1)
            k = int(np.argmax(dr)); kx = int(np.argmax(dx))
            col = cell_x_of(r["Position"][i1][:, AXIS], s.SimKernel.H_inv)
            print(f"step {m} t {pr.total_time:.12e}/{float(dd['t']):.12e} cuts {dd['cuts'].tolist()}  rho max {dr.max()/1000:.2e} at ID {r['ID'][i1][k]} type {r['Type'][i1][k]} col {col[k]} x {r['Position'][i1][k]} | pos max {dx.max():.2e} at ID {r['ID'][i1][kx]} type {r['Type'][i1][kx]} col {col[kx]} x {r['Position'][i1][kx]}  n(dr>1e-11)={int((dr/1000>1e-11).sum())}", flush=True)
            if os.environ.get("DUMP") and dr.max() / 1000 > 1e-11:
                rk = np.repeat(np.arange(len(dd["cnt"])), dd["cnt"])[i2]
                prev = globals().get("PREV_RK")
                dv = np.abs(dd["Velocity"][i2] - r["Velocity"][i1]).max(axis=1)
                for q in np.argsort(-dr)[:int(os.environ["DUMP"])]:
                    print(f"   ID {r['ID'][i1][q]:6d} type {r['Type'][i1][q]} x {r['Position'][i1][q]} col {col[q]} rank {rk[q]} (before {prev[q] if prev is not None else -1})  drho {dr[q]:.3e} dx {dx[q]:.3e} dv {dv[q]:.3e}")
            PREV_RK = np.repeat(np.arange(len(dd["cnt"])), dd["cnt"])[i2]
