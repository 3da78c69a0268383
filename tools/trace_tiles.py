#!/usr/bin/env python3
"""Experiment: occupancy over time of one neighbour-kernel launch from the per-tile start/end clocks of a
-DSPHMI_STATS build (SPHMI_TRACE_FILE).  usage: python tools/trace_tiles.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd import build  # noqa: E402

lib = "/tmp/libsphmi_stats.so"
build.build(force=True, extra_flags=["-DSPHMI_TRACE"] + [a for a in sys.argv[1:] if a.startswith("-D")], out=lib)
bench_extra = [a for a in sys.argv[1:] if not a.startswith("-D")]      # e.g. --dp 0.0085
fn = "/tmp/tiles.bin"
env = dict(os.environ, SPHMI_LIB=lib, SPHMI_TRACE_FILE=fn)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--precondition-ms", "0"] + bench_extra, env=env,
               capture_output=True)
raw = np.fromfile(fn, dtype=np.uint64); raw = raw[: 4 * (len(raw) // 36)].reshape(-1, 4)[:, 1:3]     # { kernel entry, scan start, pair loop end | XCD, exit }
raw = raw[raw[:, 1] > 0]
xcd = (raw[:, 1] >> np.uint64(60)).astype(np.int64)                 # the block's XCD rides in the top bits of the end clock
t = np.stack([raw[:, 0], raw[:, 1] & np.uint64((1 << 60) - 1)], axis=1).astype(np.int64)
if os.environ.get("SPHMI_TRACE_DUMP"):                       # the raw table for off-line schedule simulations (tools/tail_sim.py)
    np.savez(os.environ["SPHMI_TRACE_DUMP"], start=t[:, 0], end=t[:, 1], xcd=xcd)
t0, t1 = t[:, 0].min(), t[:, 1].max()
span = t1 - t0
print(f"tiles {len(t)}  span {span} ticks (100 MHz: {span / 100:.1f} us)  mean tile life {np.mean(t[:, 1] - t[:, 0]) / 100:.1f} us")
print("per XCD: tiles, first start / last end (% of span), busy wave-time share")
tot = (t[:, 1] - t[:, 0]).sum()
for x in range(8):
    m = xcd == x
    print(f"  XCD {x}: {m.sum():5d} tiles  start {100 * (t[m, 0].min() - t0) / span:5.1f} %  end {100 * (t[m, 1].max() - t0) / span:5.1f} %  "
          f"wave-time {100 * (t[m, 1] - t[m, 0]).sum() / tot:5.1f} %  95 % of its tiles done at {100 * (np.percentile(t[m, 1], 95) - t0) / span:5.1f} %")
edges = np.linspace(t0, t1, 21)
for k in range(20):
    a, b = edges[k], edges[k + 1]
    ov = np.clip(np.minimum(t[:, 1], b) - np.maximum(t[:, 0], a), 0, None).sum() / (b - a)
    print(f"  {100 * k / 20:5.1f}-{100 * (k + 1) / 20:5.1f} %  mean resident waves {ov:8.1f}  ({ov / 1024:.2f} per SIMD)")
st = (t[:, 0] - t0) / span * 100
en = (t[:, 1] - t0) / span * 100
life = (t[:, 1] - t[:, 0]) / 100.0
print("start-time histogram (% of span):", np.histogram(st, bins=[0, 5, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])[0])
print("life (us) percentiles 5/50/95/max:", np.percentile(life, [5, 50, 95, 100]).round(1))
idx = np.argsort(-en)[:15]
_r = np.fromfile(fn, dtype=np.uint64); ids = np.nonzero(_r[: 4 * (len(_r) // 36)].reshape(-1, 4)[:, 2] > 0)[0]
for i in idx:
    print(f"  tile {ids[i]:6d}  start {st[i]:5.1f} %  end {en[i]:5.1f} %  life {life[i]:6.1f} us")
for lo, hi in ((0, 30), (30, 60), (60, 80), (80, 100)):
    m = (st >= lo) & (st < hi)
    print(f"  tiles started in {lo}-{hi} %: {m.sum():6d}  mean life {life[m].mean() if m.any() else 0:6.1f} us")
