#!/usr/bin/env python3
"""Fixed cost of one sphmi_advance call (≙ one SimulationLoop call: it re-arms Δx, so its first step rebuilds the cell list):
wall time of calls of K steps on the bench workload, fitted as a + b·K, with the phase timers of the calls."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from sphexample_amd.cases import setup_dam_break_3d  # noqa: E402
from sphexample_amd.engine import make_generated_dam_break_engine  # noqa: E402

dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.00425
e = make_generated_dam_break_engine(dp, setup_dam_break_3d(dp), device_float_bytes=4)
e.advance(1e9, max_steps=10)
for K in (1, 2, 5, 10, 20, 20, 50, 100):
    t0 = time.perf_counter()
    pr = e.advance(1e9, max_steps=K)
    dt = time.perf_counter() - t0
    print(f"K={K:4d}  {dt * 1e3:8.3f} ms  ({dt / K * 1e3:.3f} ms/step)  rebuilds so far {pr.n_rebuilds}")
for name, (sec, calls) in e.timers().items():
    if calls:
        print(f"    {name:44s} {sec * 1e3:9.2f} ms  {calls:6d} calls  {sec / calls * 1e6:9.1f} us/call")
