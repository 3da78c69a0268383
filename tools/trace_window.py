#!/usr/bin/env python3
"""A window of a rocprofv3 --kernel-trace database, launch by launch: start (µs since the window's first launch), duration, the gap to
the previous launch's end, and the kernel — what a step and a cell-list rebuild of a small case look like on the device.
usage: python tools/trace_window.py <db> [first launch to show containing this name = k_cell_count] [launches before] [launches after]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_cell_count"
before, after = (int(sys.argv[3]) if len(sys.argv) > 3 else 8), (int(sys.argv[4]) if len(sys.argv) > 4 else 24)
occurrence = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rows = db.execute("select name, start, end from kernels order by start").fetchall()
hits = [i for i, r in enumerate(rows) if anchor in r[0]]
if not hits:
    raise SystemExit(f"no launch of {anchor}")
i0 = hits[min(occurrence, len(hits) - 1)]
lo, hi = max(0, i0 - before), min(len(rows), i0 + after)
t0 = rows[lo][1]
prev_end = None
for name, s, e in rows[lo:hi]:
    short = name.split("(")[0].replace("void sphmi::", "")[:70]
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:>7s}  {short}")
    prev_end = e
# per-kernel totals
print()
tot = {}
for name, s, e in rows:
    short = name.split("(")[0].replace("void sphmi::", "")[:70]
    a = tot.setdefault(short, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
span = (rows[-1][2] - rows[0][1]) / 1e3
print(f"span {span:.0f} us, kernels busy {sum(v[1] for v in tot.values()):.0f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{v[0]:6d} x {v[1] / v[0]:7.1f} us = {v[1]:9.0f} us  {k}")
