#!/usr/bin/env python3
"""Throughput in developed flow ("also report a window at t ≈ 0.4 s"): run the 1.06 M-particle dam break
to t_target, then time a window.  usage: python tools/bench_developed.py [t_target] [window_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
t_target = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
win = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dp = 0.00425
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
e = make_engine(p, s, device_float_bytes=4)
t0 = time.perf_counter(); pr = e.advance(t_target); t1 = time.perf_counter()
print(f"to t={pr.total_time:.4f}s: {pr.iteration} steps, {pr.n_rebuilds} rebuilds, {t1 - t0:.2f} s wall, "
      f"{len(p) * pr.iteration / (t1 - t0):.3g} updates/s overall")
r0 = pr.n_rebuilds
e.force_kernel_stats(reset=True)
t0 = time.perf_counter(); pr = e.advance(1e9, max_steps=win); t1 = time.perf_counter()
ms, n = e.force_kernel_stats()
print(f"window of {win} steps at t≈{pr.total_time:.3f}s: {len(p) * win / (t1 - t0):.3g} updates/s, {1e3 * (t1 - t0) / win:.3f} ms/step, "
      f"{pr.n_rebuilds - r0} rebuilds, kernel {ms:.3f} ms/launch")
for name, (sec, calls) in e.timers().items():
    if calls:
        print(f"    {name:46s} {sec:8.3f} s  {calls:7d} calls")
