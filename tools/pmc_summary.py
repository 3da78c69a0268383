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
    for k, d, c, v in rows:
        if kernel_filter and kernel_filter not in k:
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
