#!/usr/bin/env python3
"""Loop counters (-DSPHMI_STATS) and per-tile clocks (-DSPHMI_TRACE) of a small case with PREBUILT variant libraries
(tools/prebuild_variants.py; nothing is compiled on the GPU box).
usage: python tools/small_case_probe.py LIB CASE FLOAT_BYTES [steps]        LIB: path of a -DSPHMI_STATS and / or -DSPHMI_TRACE build"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--child":
    import conftest
    from sphexample_amd.engine import make_engine
    p, s = getattr(conftest, "load_" + sys.argv[2])()
    e = make_engine(p, s, device_float_bytes=int(sys.argv[3]))
    if hasattr(p, "geometries"):
        e.set_motions(p.geometries)
    e.advance(1e9, max_steps=int(sys.argv[4]))
    del e
    sys.exit(0)
lib, case, fb = sys.argv[1], sys.argv[2], sys.argv[3]
steps = sys.argv[4] if len(sys.argv) > 4 else "60"
fn = "/tmp/tiles_probe.bin"
if os.path.exists(fn):
    os.remove(fn)
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case, fb, steps], env=dict(os.environ, SPHMI_LIB=lib, SPHMI_TRACE_FILE=fn),
                   capture_output=True, text=True)
print(f"## {os.path.basename(lib)} {case} fp{int(fb) * 8}")
for l in r.stderr.splitlines():
    if "stats" in l:
        w = l.replace("[sphmi stats]", "").split()
        v = {" ".join(w[i:i + 1]): 0 for i in range(0)}
        nums = [int(x) for x in w if x.isdigit()]
        it, lane, ref, emp, ch, waves = nums[:6]
        print(f"  waves {waves}  pair-loop iterations per wave {it / max(waves, 1):.1f}  lanes busy {lane / max(64 * it, 1):.1%}  chunks per wave {ch / max(waves, 1):.2f}")
if r.returncode:
    print(r.stderr[-2000:])
if os.path.exists(fn):
    raw = np.fromfile(fn, dtype=np.uint64)
    raw = raw[: 4 * (len(raw) // 36)].reshape(-1, 4)
    raw = raw[raw[:, 2] > 0]
    if len(raw):
        t = raw.astype(np.int64)
        t[:, 2] = (raw[:, 2] & np.uint64((1 << 60) - 1)).astype(np.int64)
        us = (t - t[:, 0].min()) / 100.0
        med = lambda a: float(np.median(a))  # noqa: E731
        print(f"  tiles {len(t)} (last corrector launch), us after the first wave's entry")
        print(f"  entry        median {med(us[:, 0]):5.2f}  last {us[:, 0].max():5.2f}")
        print(f"  scan start   median {med(us[:, 1]):5.2f}  last {us[:, 1].max():5.2f}   prologue   median {med(us[:, 1] - us[:, 0]):5.2f}")
        print(f"  pairs done   median {med(us[:, 2]):5.2f}  last {us[:, 2].max():5.2f}   scan+pairs median {med(us[:, 2] - us[:, 1]):5.2f}  max {(us[:, 2] - us[:, 1]).max():5.2f}")
        print(f"  exit         median {med(us[:, 3]):5.2f}  last {us[:, 3].max():5.2f}   epilogue   median {med(us[:, 3] - us[:, 2]):5.2f}")
