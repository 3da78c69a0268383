#!/usr/bin/env python3
"""Two stages for profiling developed flow: `save T FILE` runs the 1.06 M dam break to time T and stores the particle
state; `run FILE STEPS` reloads it into a fresh engine and advances STEPS steps (put this one under rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
from sphexample_amd.preprocess import particles_from_arrays
dp = 0.00425
s = setup_dam_break_3d(dp)
if sys.argv[1] == "save":
    e = make_engine(dam_break_3d(dp), s, device_float_bytes=4)
    pr = e.advance(float(sys.argv[2]))
    d = e.download()
    np.savez(sys.argv[3], **d)
    print("saved", pr.total_time, pr.iteration)
else:
    d = dict(np.load(sys.argv[2]))
    p = particles_from_arrays(3, d["Position"], d["Density"], d["Type"], d["GroupMarker"], d["ID"], sort_by_id=False)
    p.Velocity[:] = d["Velocity"]; p.Acceleration[:] = d["Acceleration"]
    e = make_engine(p, s, device_float_bytes=4)
    e.advance(1e9, max_steps=3)
    e.force_kernel_stats(reset=True)
    t0 = time.perf_counter(); e.advance(1e9, max_steps=int(sys.argv[3])); t1 = time.perf_counter()
    print("steps", sys.argv[3], "ms/step", 1e3 * (t1 - t0) / int(sys.argv[3]), "kernel", e.force_kernel_stats())
