#!/usr/bin/env python3
"""Throughput of fp64 handles (the reference's examples are Float64) next to fp32 on the 3-D dam break."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.00425
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
for fb in (4, 8):
    e = make_engine(p, s, device_float_bytes=fb)
    e.advance(1e9, max_steps=5)
    t0 = time.perf_counter(); e.advance(1e9, max_steps=steps); dt = time.perf_counter() - t0
    ms, n = e.force_kernel_stats()
    print(f"fp{fb * 8}: N={len(p)}  {dt / steps * 1e3:.3f} ms/step  {len(p) * steps / dt:.3g} updates/s  kernel {ms:.3f} ms/launch")
