#!/usr/bin/env python3
"""Per-kernel sums of PMC counters from a rocprofv3 rocpd database (counter collection run)."""
import sqlite3
import sys
from collections import defaultdict


def main(path, kernel_filter=""):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(f"select {name_col}, dispatch_id, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    # per (kernel, dispatch, counter) totals first: dispatches of cancelled steps (queued, then told by the device-side
    # step control to return at once) show ≈0 in every counter and are left out of the averages
    per = defaultdict(float)
    for k, d, c, v in rows:
        if kernel_filter and kernel_filter not in k:
            continue
        per[(k, d, c)] += v
    peak = defaultdict(float)
    for (k, d, c), v in per.items():
        peak[(k, c)] = max(peak[(k, c)], v)
    dead = {(k, d) for (k, d, c), v in per.items() if peak[(k, c)] > 0 and v < 0.05 * peak[(k, c)]}
    for (k, d, c), v in per.items():
        if (k, d) in dead:
            continue
        agg[k][c] += v
        disp[k].add(d)
    for k in agg:
        n = len(disp[k])
        print(f"## {k[:100]}  ({n} dispatches; per-dispatch averages)")
        for c in sorted(agg[k]):
            print(f"  {c:32s} {agg[k][c] / n:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
