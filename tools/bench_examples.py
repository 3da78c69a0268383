#!/usr/bin/env python3
"""Step time and phase timers of the reference's example cases (tests/golden/input layouts) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
from sphexample_amd.engine import make_engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name in ("dam_break_2d", "dam_break_2d_mdbc", "still_wedge", "still_wedge_middle_square", "moving_square", "duckling", "dam_break_3d_shipped"):
    p, s = getattr(conftest, "load_" + name)()
    for fb in (4, 8):
        e = make_engine(p, s, device_float_bytes=fb)
        if hasattr(p, "geometries"):
            e.set_motions(p.geometries)
        e.advance(1e9, max_steps=20)
        t0 = time.perf_counter(); pr = e.advance(1e9, max_steps=steps); dt = time.perf_counter() - t0
        tm = {k.split()[0]: v[0] / max(v[1], 1) * 1e6 for k, v in e.timers().items() if v[1]}
        print(f"{name:26s} fp{fb * 8} N={len(p):6d}  {dt / steps * 1e6:7.1f} us/step  {len(p) * steps / dt:.3g} upd/s  rebuilds {pr.n_rebuilds:4d}  "
              + "  ".join(f"{k}:{v:.0f}us" for k, v in tm.items()), flush=True)
