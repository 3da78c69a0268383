#!/usr/bin/env python3
"""Waves per tile against the size of the launch: µs per step of the 3-D dam-break lattice for $SPHMI_WPT = 1, 2, 4, 8 at sizes
around the thresholds of Engine::launch_force_model (kWptTiny / kWptSmall / kWptMedium tiles).   python tools/wpt_sweep.py [fb] [laminar]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fb = sys.argv[1] if len(sys.argv) > 1 else "4"
model = sys.argv[2] if len(sys.argv) > 2 else "default"
code = r'''
import sys, time
sys.path.insert(0, %r)
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
dp, fb = float(sys.argv[1]), int(sys.argv[2])
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
if sys.argv[3] == 'laminar':
    import dataclasses
    from sphexample_amd import Laminar
    s = dataclasses.replace(s, SimViscosity=Laminar())
e = make_engine(p, s, device_float_bytes=fb)
e.advance(1e9, max_steps=30)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); e.advance(1e9, max_steps=200); best = min(best, time.perf_counter() - t0)
print(len(p), best / 200 * 1e6)
''' % ROOT
print("dp       N      tiles |  wpt=auto     1       2       4       8   (us/step)")
for dp in ("0.02", "0.016", "0.0135", "0.0115", "0.0100", "0.0085", "0.0075", "0.0067", "0.0057"):
    row, n = [], 0
    for w in ("0", "1", "2", "4", "8"):
        env = dict(os.environ)
        if w != "0":
            env["SPHMI_WPT"] = w
        else:
            env.pop("SPHMI_WPT", None)
        r = subprocess.run([sys.executable, "-c", code, dp, fb, model], env=env, capture_output=True, text=True)
        try:
            n, us = r.stdout.split(); row.append(f"{float(us):7.1f}")
        except Exception:
            row.append("   fail")
    print(f"{dp:8s} {int(n):7d} {(int(n) + 63) // 64:6d} | " + " ".join(row), flush=True)
