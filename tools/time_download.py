#!/usr/bin/env python3
"""Cost of one output cycle (sphmi_download of every field) next to the compute it interrupts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
dp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.00425
p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
e = make_engine(p, s, device_float_bytes=4)
e.advance(1e9, max_steps=5)
t0 = time.perf_counter(); pr = e.advance(1e9, max_steps=25); t1 = time.perf_counter()
d = e.download(); t2 = time.perf_counter()
d = e.download(("Position", "Velocity", "Density", "Pressure", "ID", "Type")); t3 = time.perf_counter()
print(f"N={len(p)}: 25 steps {1e3*(t1-t0):.1f} ms; download(all fields) {1e3*(t2-t1):.1f} ms; download(6 fields) {1e3*(t3-t2):.1f} ms")
q = p.copy()
for k in range(4):
    if k == 2:
        e.pin(q); print('pinned')
    t = time.perf_counter(); e.download_into(q); print(f"download_into #{k}: {1e3 * (time.perf_counter() - t):.1f} ms")
# asynchronous output: begin → advance → end
t = time.perf_counter(); e.download_into_begin(q); tb = time.perf_counter() - t
t = time.perf_counter(); e.advance(1e9, max_steps=25); ta = time.perf_counter() - t
t = time.perf_counter(); e.download_end(); te = time.perf_counter() - t
print(f"async: begin {1e3 * tb:.2f} ms, 25 steps {1e3 * ta:.1f} ms, end {1e3 * te:.2f} ms")
