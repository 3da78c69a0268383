#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into the per-kernel stats table that
`rocprofv3 --stats` prints: calls, total / average / min / max duration, share.  Usage:
    python tools/prof_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    # Steps are queued in batches of up to 16 with the control on the device (k_step_control): the pass kernels of a step
    # the control cancelled (rebuild pending, loop bound reached) return at once.  Those launches (< 5 % of the
    # kernel's longest one) are counted separately so that the averages describe the launches that did the work.
    rows = db.execute(
        "select k.name, count(*), sum(k.duration), avg(k.duration), min(k.duration), max(k.duration), "
        "max(k.vgpr_count), max(k.sgpr_count), max(k.lds_size), max(k.scratch_size), max(k.grid_x), max(k.workgroup_x) "
        "from kernels k join (select name, max(duration) as mx from kernels group by name) m on m.name = k.name "
        "where k.duration >= 0.05 * m.mx group by k.name order by sum(k.duration) desc").fetchall()
    cancelled = dict(db.execute(
        "select k.name, count(*) from kernels k join (select name, max(duration) as mx from kernels group by name) m "
        "on m.name = k.name where k.duration < 0.05 * m.mx group by k.name").fetchall())
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        print(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | "
              f"{100 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
    if cancelled:
        print()
        print("Cancelled launches (returned at once, excluded above): " +
              ", ".join(f"`{k[:60]}` × {v}" for k, v in cancelled.items()))


if __name__ == "__main__":
    main(sys.argv[1])
