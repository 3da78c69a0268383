#!/usr/bin/env python3
"""How well does the work measure (candidates per particle) predict a rank's kernel time?  Cut the generated dam break
of `world` ranks' bench size into its slabs, run every slab (with one ghost column per side as ordinary particles) as a
stand-alone engine on the one GPU and time the neighbour kernel.  usage: python tools/slab_balance_probe.py world [count]
("count": cut by particle count along x instead — the first version of the slab plan)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))      # slab_planner_reference: the numpy planner the tests hold the C++ one to
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from slab_planner_reference import SlabPlan, cell_x_of, choose_axis, particle_work
from sphexample_amd.engine import make_engine

world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
by_count = len(sys.argv) > 2 and sys.argv[2] == "count"
dp = 0.00425 / world ** (1.0 / 3.0)
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
cols = [cell_x_of(p.Position[:, a].astype(np.float32).astype(np.float64), s.SimKernel.H_inv) for a in range(3)]
w = particle_work(cols)
if by_count:
    ax = choose_axis(cols, world, 2, None); plan = SlabPlan.from_columns(cols[ax], world, 2, None)
else:
    ax = choose_axis(cols, world, 2, w); plan = SlabPlan.from_columns(cols[ax], world, 2, w)
cx = cols[ax]
print(f"world {world} dp {dp:.6f} N {len(p)} axis {ax} cuts {plan.cuts()} ({'count' if by_count else 'work'}-balanced)")
times, works, counts = [], [], []
for r in range(world):
    lo, hi = plan.cx_lo[r], plan.cx_hi[r]
    own = (cx >= lo) & (cx <= hi)
    sel = (cx >= lo - 1) & (cx <= hi + 1)
    q = p.copy(); q.permute(np.nonzero(sel)[0])
    e = make_engine(q, s, device_float_bytes=4)
    e.advance(1e9, max_steps=10)
    e.force_kernel_stats(reset=True)
    e.advance(1e9, max_steps=40)
    ms, n = e.force_kernel_stats()
    times.append(ms); works.append(w[own].sum()); counts.append(int(own.sum()))
    print(f"  rank {r}: owned {counts[-1]:8d} (+ghost {int(sel.sum()) - counts[-1]:7d})  work {works[-1]:.4g}  kernel {ms:.4f} ms/launch", flush=True)
    del e
t, wk, c = np.array(times), np.array(works, dtype=float), np.array(counts, dtype=float)
print(f"imbalance max/mean: kernel time {t.max() / t.mean():.3f}   work {wk.max() / wk.mean():.3f}   count {c.max() / c.mean():.3f}")
