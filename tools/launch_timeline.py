#!/usr/bin/env python3
"""Per-launch timeline of a rocprofv3 --kernel-trace database: start (ms since the first kernel), duration and name of every
kernel longer than `min_us`, in launch order — what the first steps of a short bench window look like."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
for name, s, e in rows:
    d = (e - s) / 1e3
    if d >= min_us:
        short = name.split("(")[0].replace("void sphmi::", "")[:60]
        print(f"{(s - t0) / 1e6:9.3f} ms  {d:8.1f} us  {short}")
