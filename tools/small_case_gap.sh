# Round 5: how much of a small handle's step is NOT kernel execution?  rocprofv3 kernel trace of 2000 steps of an example layout:
# sum of kernel durations per step against the wall time per step.   usage: tools/small_case_gap.sh <case> [fb]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; c=${1:-dam_break_2d}; fb=${2:-4}
out=$R/gpurun_out/gap_$c; mkdir -p $out
cat > /tmp/gap.py <<PY
import sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import conftest
from sphexample_amd.engine import make_engine
p, s = getattr(conftest, "load_$c")()
e = make_engine(p, s, device_float_bytes=$fb)
if hasattr(p, "geometries"): e.set_motions(p.geometries)
e.advance(1e9, max_steps=200)
t0 = time.perf_counter(); e.advance(1e9, max_steps=2000); dt = time.perf_counter() - t0
print(f"WALL $c fp{8*$fb}: {dt / 2000 * 1e6:.2f} us per step")
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python /tmp/gap.py > $out/log.txt 2>&1
grep WALL $out/log.txt
db=$(find $out/trace -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" | head -8
python - <<PY
import sqlite3
db = sqlite3.connect("$db")
rows = db.execute("select start, end, name from kernels order by start").fetchall()
rows = rows[len(rows) // 3:]            # the timed 2000 steps (the tail of the trace)
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
gaps.sort()
print(f"kernels {len(rows)}: busy {busy / 1e3:.0f} us of a span of {span / 1e3:.0f} us = {100 * busy / span:.1f} %; gap between consecutive kernels: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us")
PY
rm -rf $out/trace
