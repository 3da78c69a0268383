#!/usr/bin/env python3
"""Where a small case's pass goes: per-tile start / end clocks (-DSPHMI_TRACE build, 100 MHz s_memrealtime) of the LAST
corrector launch of the 2-D dam break (6 881 particles, 108 tiles × 4 waves), against the kernel duration rocprof reports.
usage (GPU box): python tools/trace_small.py [float bytes]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import conftest
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + os.environ.get("TRACE_CASE", "dam_break_2d"))()
    e = make_engine(p, s, device_float_bytes=int(sys.argv[2]))
    e.advance(1e9, max_steps=60)
    del e
    sys.exit(0)

from sphexample_amd import build  # noqa: E402
lib, fn = "/tmp/libsphmi_trace.so", "/tmp/tiles_small.bin"
build.build(force=True, extra_flags=["-DSPHMI_TRACE"] + [a for a in sys.argv[2:] if a.startswith("-D")], out=lib)
fb = sys.argv[1] if len(sys.argv) > 1 else "4"
subprocess.run([sys.executable, os.path.abspath(__file__), "--child", fb], env=dict(os.environ, SPHMI_LIB=lib, SPHMI_TRACE_FILE=fn), check=True)
raw = np.fromfile(fn, dtype=np.uint64)
raw = raw[: 4 * (len(raw) // 36)].reshape(-1, 4)           # per tile: kernel entry, scan start, pair loop end | XCD << 60, exit
raw = raw[raw[:, 2] > 0]
t = raw.astype(np.int64)
t[:, 2] = (raw[:, 2] & np.uint64((1 << 60) - 1)).astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
med = lambda a: float(np.median(a))  # noqa: E731
print(f"tiles {len(t)} (last corrector launch); times in us after the first wave's entry")
print(f"  entry        median {med(us[:, 0]):5.2f}  last {us[:, 0].max():5.2f}")
print(f"  scan start   median {med(us[:, 1]):5.2f}  last {us[:, 1].max():5.2f}   prologue   median {med(us[:, 1] - us[:, 0]):5.2f}")
print(f"  pairs done   median {med(us[:, 2]):5.2f}  last {us[:, 2].max():5.2f}   scan+pairs median {med(us[:, 2] - us[:, 1]):5.2f}")
print(f"  exit         median {med(us[:, 3]):5.2f}  last {us[:, 3].max():5.2f}   epilogue   median {med(us[:, 3] - us[:, 2]):5.2f}")
