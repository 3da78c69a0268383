import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import conftest, numpy as np
from sphexample_amd.engine import make_engine
for name, T in (("dam_break_2d", 1.5), ("dam_break_2d_mdbc", 1.0), ("moving_square", 0.5), ("duckling", 0.3), ("dam_break_3d_shipped", 1.0)):
    p, s = getattr(conftest, "load_" + name)()
    e = make_engine(p, s, device_float_bytes=4)
    if hasattr(p, "geometries"): e.set_motions(p.geometries)
    t0 = time.perf_counter(); pr = e.advance(T); dt = time.perf_counter() - t0
    d = e.download(("Density", "Position"))
    print(f"{name:22s} t={pr.total_time:.3f} steps {pr.iteration:6d} rebuilds {pr.n_rebuilds:5d} wall {dt:6.2f}s  {len(p) * pr.iteration / dt:.3g} upd/s  rho [{d['Density'].min():.1f}, {d['Density'].max():.1f}]  nan {int(np.isnan(d['Position']).sum())}", flush=True)
