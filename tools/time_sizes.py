#!/usr/bin/env python3
"""µs per step of the generated 3-D dam break at a list of spacings (from rest, 100 steps after 20), fp32 and fp64.
usage: python tools/time_sizes.py DP [DP ...]      ($SPHMI_LIB: another build)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
for dp in [float(x) for x in sys.argv[1:]]:
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    for fb in (4, 8):
        best = 1e9
        for rep in range(2):
            e = make_engine(p, s, device_float_bytes=fb)
            e.advance(1e9, max_steps=20)
            t0 = time.perf_counter(); e.advance(1e9, max_steps=100); dt = time.perf_counter() - t0
            best = min(best, dt)
            del e
        print(f"dp {dp} N={len(p)} tiles={(len(p) + 63) // 64} fp{fb * 8}: {best / 100 * 1e6:8.1f} us/step  {len(p) * 100 / best:.3e} upd/s", flush=True)
