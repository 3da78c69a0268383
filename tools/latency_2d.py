#!/usr/bin/env python3
"""Step latency of the small 2-D cases (C1/C2 of SURVEY.md §8d) on one GPU: µs per step and the phase timers."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from sphexample_amd.cases import setup_dam_break_2d  # noqa: E402
from sphexample_amd.engine import make_engine  # noqa: E402
from sphexample_amd.preprocess import AllocateDataStructures  # noqa: E402
from sphexample_amd import Geometry, Fixed, Fluid  # noqa: E402

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "input")
geo = [Geometry(os.path.join(root, "DamBreak2d_Dp0.02_Bound.csv"), 1, Fixed, None, 2, "Float64"),
       Geometry(os.path.join(root, "DamBreak2d_Dp0.02_Fluid.csv"), 2, Fluid, None, 2, "Float64")]
p = AllocateDataStructures(geo)
s = setup_dam_break_2d()
for fb in (4, 8):
    e = make_engine(p, s, device_float_bytes=fb)
    e.advance(1e9, max_steps=50)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    t0 = time.perf_counter()
    pr = e.advance(1e9, max_steps=steps)
    dt = time.perf_counter() - t0
    print(f"fp{fb * 8}: N={len(p)}  {dt / steps * 1e6:.1f} us/step  {len(p) * steps / dt:.3g} updates/s  rebuilds={pr.n_rebuilds}")
    for name, (sec, calls) in e.timers().items():
        if calls:
            print(f"    {name:28s} {sec * 1e3:9.2f} ms  {calls:6d} calls  {sec / calls * 1e6:8.1f} us/call")
