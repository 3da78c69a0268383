#!/usr/bin/env python3
"""Where a step of the slab driver goes, from a rocprofv3 --kernel-trace database of a bench.py run.

  python tools/slab_trace.py <db> <steps in the trace = warm-up + timed> [label]

Per kernel class: launches per step, µs per launch, µs per step; and the timeline: wall span per step, time with at least one kernel
running (union), summed kernel time (≥ union when launches of different streams overlap), idle gaps.  The neighbour kernel is split by
grid size into the INTERIOR launch (the big one) and the EDGE launch of a pass."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2])
label = sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else None
rows = db.execute(f"select name, start, end{', ' + gx if gx else ''} from kernels order by start").fetchall()
# the stepping part: from the first neighbour-force launch to the last
nf = [i for i, r in enumerate(rows) if "k_neighbor_force" in r[0]]
rows = rows[nf[0]:nf[-1] + 1]
# drop everything before the measured handle's first step (scratch handles, generators): keep the last `steps` steps' worth of launches,
# found through the per-step control kernel when there is one
ctl = [i for i, r in enumerate(rows) if "k_dd_merge_control" in r[0] or "k_step_control" in r[0]]


def short(n):
    n = n.split("(")[0].replace("void sphmi::", "").replace("sphmi::", "")
    return re.sub(r"<.*", "", n)


grids = sorted({r[3] for r in rows if "k_neighbor_force" in r[0]}) if gx else []
big = max(grids) if grids else 0
cls = {}
for r in rows:
    k = short(r[0])
    if k == "k_neighbor_force":
        pas = "pass 1" if re.search(r"<\w+, \d, 1,", r[0]) else ("pass 2" if re.search(r"<\w+, \d, 2,", r[0]) else "forces")
        k = f"k_neighbor_force {pas} " + ("interior" if not gx or r[3] >= 0.5 * big else "edge")
        if (r[2] - r[1]) < 3000:
            k += " (cancelled)"
    a = cls.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
span = (rows[-1][2] - rows[0][1]) / 1e3
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
busy, depth, last = 0.0, 0, None
for t, d in ev:
    if depth > 0:
        busy += (t - last) / 1e3
    depth += d; last = t
total = sum(v[1] for v in cls.values())
print(f"### {label}: {len(rows)} launches over {steps} steps")
print()
print("| kernel | launches / step | µs / launch | µs / step |")
print("|---|---|---|---|")
for k, v in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {v[0] / steps:.2f} | {v[1] / v[0]:.1f} | {v[1] / steps:.1f} |")
print()
print(f"wall span per step {span / steps:.1f} µs; at least one kernel running {busy / steps:.1f} µs; summed kernel time {total / steps:.1f} µs "
      f"(overlap {max(total - busy, 0) / steps:.1f} µs); idle {max(span - busy, 0) / steps:.1f} µs per step")
