#!/usr/bin/env python3
"""µs per step and phase timers of ONE example layout: python tools/steps_case.py CASE FLOAT_BYTES STEPS [repetitions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
from sphexample_amd.engine import make_engine
name, fb, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
p, s = getattr(conftest, "load_" + name)()
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
    e = make_engine(p, s, device_float_bytes=fb)
    if hasattr(p, "geometries"):
        e.set_motions(p.geometries)
    e.advance(1e9, max_steps=20)
    t0 = time.perf_counter(); pr = e.advance(1e9, max_steps=steps); dt = time.perf_counter() - t0
    tm = {k.split()[0]: v[0] / max(v[1], 1) * 1e6 for k, v in e.timers().items() if v[1]}
    print(f"{name:26s} fp{fb * 8} N={len(p):6d}  {dt / steps * 1e6:7.1f} us/step  rebuilds {pr.n_rebuilds:4d}  " + "  ".join(f"{k}:{v:.0f}us" for k, v in tm.items()), flush=True)
    del e
