#!/bin/bash
# Every kernel instantiation of tools/bench_variants.py in this tree and in the previous round's (build/r3tree = `git archive <commit> | tar -x`,
# library built in place, tools/bench_variants.py copied in), one gpurun call, then the table.
# The next round's reference: mkdir -p build/r4tree && git archive <last commit of round 4> | tar -x -C build/r4tree && (cd build/r4tree && python -m sphexample_amd.build --force)   usage (gpurun): bash tools/variants_vs_previous.sh <prev tree> <out md>
cd $GRAFT_REPO_ROOT
prev=${1:-build/r3tree}; out=${2:-gpurun_out/variants_vs_previous.md}
# (round 5: the first instantiations of the FIRST process ran on a device that had just left its idle state — +8 … +14 % on the 17 k-particle
# run-time-model kernels of "this tree", which a targeted A/B, tools/variant_probe.py, did not reproduce: both trees now start behind a warm-up)
python tools/variant_probe.py 0.0085 default 4 3000 > /dev/null 2>&1
python tools/bench_variants.py 200 > gpurun_out/variants_head.txt 2>/dev/null
python tools/variant_probe.py 0.0085 default 4 3000 > /dev/null 2>&1
(cd $prev && python tools/bench_variants.py 200 > $GRAFT_REPO_ROOT/gpurun_out/variants_prev.txt 2>/dev/null)
python - "$out" <<'P'
import math, re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"dp (\S+)\s+N=\s*(\d+) (\S+)\s+(fp\d+)\s+(one slab|\d slabs)\s+([\d.]+) us/step", l)
        if m: d[(m[1], m[2], m[3], m[4], m[5])] = float(m[6])
    return d
a, b = load("gpurun_out/variants_head.txt"), load("gpurun_out/variants_prev.txt")
keys = [k for k in a if k in b]
g = math.exp(sum(math.log(a[k] / b[k]) for k in keys) / len(keys))
with open(sys.argv[1], "w") as f:
    f.write(f"geometric mean of the step-time ratios: {g:.3f} over {len(keys)} instantiations (200 steps from rest after 20)\n\n")
    f.write("| instantiation | this tree µs/step | previous round µs/step (same box, same call) | ratio |\n|---|---|---|---|\n")
    for k in keys:
        f.write(f"| dp {k[0]} N={k[1]} {k[2]} {k[3]} {k[4]} | {a[k]:.1f} | {b[k]:.1f} | {a[k] / b[k]:.3f} |\n")
print(open(sys.argv[1]).read()[:400])
worst = sorted(keys, key=lambda k: -a[k] / b[k])[:6]
print("slowest ratios:", [(k, round(a[k] / b[k], 3)) for k in worst])
P
