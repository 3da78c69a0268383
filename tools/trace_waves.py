#!/usr/bin/env python3
"""Experiment: where every WAVE of one neighbour-kernel launch ran (-DSPHMI_TRACE -DSPHMI_TRACE_WAVES: start / end of its scan +
pair loop, HW_ID, work counters) — load per SIMD and per compute unit, and how the waves of a tile compare.
usage (GPU box): python tools/trace_waves.py [--dp 0.0085] [-DFLAG …]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd import build  # noqa: E402

lib, fn = "/tmp/libsphmi_tw.so", "/tmp/waves.bin"
build.build(force=True, extra_flags=["-DSPHMI_TRACE", "-DSPHMI_TRACE_WAVES"] + [a for a in sys.argv[1:] if a.startswith("-D")], out=lib)
extra = [a for a in sys.argv[1:] if not a.startswith("-D")]
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--precondition-ms", "0"] + extra,
               env=dict(os.environ, SPHMI_LIB=lib, SPHMI_TRACE_FILE=fn), capture_output=True)
raw = np.fromfile(fn, dtype=np.uint64)
nt = len(raw) // 36
W = raw[4 * nt:].reshape(nt, 8, 4)
tile, wv = np.nonzero(W[:, :, 1] > 0)
st, en = W[tile, wv, 0].astype(np.int64), W[tile, wv, 1].astype(np.int64)
hw, xcd = (W[tile, wv, 2] & np.uint64(0xffffffff)).astype(np.int64), (W[tile, wv, 2] >> np.uint64(32)).astype(np.int64)
it, ch = (W[tile, wv, 3] & np.uint64(0xffffffff)).astype(np.int64), (W[tile, wv, 3] >> np.uint64(32)).astype(np.int64)
work = 9 * it + 16 * ch                                   # the kernel's own work measure (≈ vector-ALU cycles / 30)
t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcd * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
print(f"waves {len(st)} of {len(np.unique(tile))} tiles; span {en.max():.1f} us; wave life median {np.median(en - st):.1f} us, max {np.max(en - st):.1f} us")
for name, key in (("SIMD", simd_key), ("compute unit", cu_key)):
    u, inv = np.unique(key, return_inverse=True)
    n = np.bincount(inv); w = np.bincount(inv, weights=work); last = np.zeros(len(u)); np.maximum.at(last, inv, en)
    print(f"per {name}: {len(u)} used; waves min/mean/max {n.min()} / {n.mean():.2f} / {n.max()}; work max/mean {w.max() / w.mean():.3f}, "
          f"cv {w.std() / w.mean():.3f}; last end min/median/max {last.min():.1f} / {np.median(last):.1f} / {last.max():.1f} us; "
          f"corr(work, last end) {np.corrcoef(w, last)[0, 1]:.2f}")
print("waves per SIMD id 0..3:", np.bincount(simd, minlength=4), " work per SIMD id:", np.bincount(simd, weights=work, minlength=4).astype(np.int64))
print("SIMD ids of wave 0 / wave 1 of a tile:", np.bincount(simd[wv == 0], minlength=4), "/", np.bincount(simd[wv == 1], minlength=4))
# the waves of a tile
nw = np.bincount(tile)
if nw.max() > 1:
    wt = np.zeros((nt, 8)); wt[tile, wv] = work
    et = np.zeros((nt, 8)); et[tile, wv] = en
    m = nw[: nt] > 1 if len(nw) >= nt else np.pad(nw, (0, nt - len(nw))) > 1
    k = int(nw.max())
    hi, lo = wt[m, :k].max(1), wt[m, :k].min(1)
    print(f"waves of a tile ({k} each): work of the heavier / lighter wave, median ratio {np.median(hi / np.maximum(lo, 1)):.2f}, "
          f"95 % {np.percentile(hi / np.maximum(lo, 1), 95):.2f}; end-time gap median {np.median(et[m, :k].max(1) - et[m, :k].min(1)):.1f} us")
    same = [(simd_key[(tile == t)][0] // 4 == simd_key[(tile == t)][1] // 4) for t in np.unique(tile)[:400]]
    print(f"both waves of a tile on the same compute unit: {100 * np.mean(same):.0f} % (first 400 tiles)")
if nw.max() > 2:
    k = int(nw.max())
    print("by wave of the tile: mean pair-loop iterations / chunks / life (us):")
    for w_ in range(k):
        m_ = wv == w_
        print(f"   wave {w_}: {it[m_].mean():6.1f} (max {it[m_].max():4d})  {ch[m_].mean():4.1f}  {np.mean(en[m_] - st[m_]):6.1f} (max {np.max(en[m_] - st[m_]):5.1f})")
    print(f"   life vs iterations: corr {np.corrcoef(it, en - st)[0, 1]:.2f}; us per iteration (fit) {np.polyfit(it, en - st, 1)[0]:.3f}, intercept {np.polyfit(it, en - st, 1)[1]:.2f} us")
# the stragglers
o = np.argsort(-en)[:10]
for i in o:
    print(f"  tile {tile[i]:6d} wave {wv[i]}  end {en[i]:6.1f} us  life {en[i] - st[i]:6.1f}  it {it[i]:4d} ch {ch[i]:2d} work {work[i]:6d} (mean {work.mean():.0f})  waves on its SIMD {np.sum(simd_key == simd_key[i])}, their work {work[simd_key == simd_key[i]].sum()}")
