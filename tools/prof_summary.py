#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into the per-kernel stats table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration, share.  Usage:
    python tools/prof_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | "
              f"{100 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main(sys.argv[1])
