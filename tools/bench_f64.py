#!/usr/bin/env python3
"""fp64 handles (the reference's own arithmetic) at three sizes, compiled-in and run-time models: µs per step for the library
named by $SPHMI_LIB.   python tools/bench_f64.py [steps]"""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd import Laminar  # noqa: E402
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d  # noqa: E402
from sphexample_amd.engine import make_engine  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = []
for dp in (0.0085, 0.0057, 0.00425):
    p, s0 = dam_break_3d(dp), setup_dam_break_3d(dp)
    for name, s in (("default", s0), ("laminar", dataclasses.replace(s0, SimViscosity=Laminar()))):
        e = make_engine(p, s, device_float_bytes=8)
        e.advance(1e9, max_steps=20)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); e.advance(1e9, max_steps=steps); best = min(best, time.perf_counter() - t0)
        out.append(f"{best / steps * 1e6:8.1f}")
print(os.path.basename(os.environ.get("SPHMI_LIB", "shipped")), " ".join(out), flush=True)
