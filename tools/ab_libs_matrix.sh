# Two prebuilt libraries over every kernel instantiation of tools/bench_variants.py and the example layouts, one box, interleaved warm-ups:
#   bash tools/ab_libs_matrix.sh <variant a> <variant b> <out prefix under gpurun_out/r05/>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
a=$1; b=$2; out=gpurun_out/r05/$3
python tools/variant_probe.py 0.0085 default 4 3000 > /dev/null 2>&1
SPHMI_LIB=$PWD/build/variants/libsphmi_$a.so python tools/bench_variants.py 200 > ${out}_$a.txt 2>/dev/null
SPHMI_LIB=$PWD/build/variants/libsphmi_$b.so python tools/bench_variants.py 200 > ${out}_$b.txt 2>/dev/null
SPHMI_LIB=$PWD/build/variants/libsphmi_$a.so python tools/bench_examples.py 2000 2>&1 | grep -v "^\[" > ${out}_examples_$a.txt
SPHMI_LIB=$PWD/build/variants/libsphmi_$b.so python tools/bench_examples.py 2000 2>&1 | grep -v "^\[" > ${out}_examples_$b.txt
python - ${out}_$a.txt ${out}_$b.txt ${out}_examples_$a.txt ${out}_examples_$b.txt <<'P'
import math, re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"dp (\S+)\s+N=\s*(\d+) (\S+)\s+(fp\d+)\s+(one slab|\d slabs)\s+([\d.]+) us/step", l)
        if m: d[(m[1], m[2], m[3], m[4], m[5])] = float(m[6])
        m = re.match(r"(\S+)\s+(fp\d+) N=\s*(\d+)\s+([\d.]+) us/step", l)
        if m: d[(m[1], m[2], m[3])] = float(m[4])
    return d
for pa, pb in ((sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])):
    a, b = load(pa), load(pb)
    keys = [k for k in a if k in b]
    g = math.exp(sum(math.log(b[k] / a[k]) for k in keys) / len(keys))
    print(f"{pb} / {pa}: geometric mean of the step-time ratios {g:.3f} over {len(keys)}")
    for k in sorted(keys, key=lambda k: -b[k] / a[k])[:8]: print("   slowest", k, a[k], b[k], f"{b[k] / a[k]:.3f}")
    for k in sorted(keys, key=lambda k: b[k] / a[k])[:5]: print("   fastest", k, a[k], b[k], f"{b[k] / a[k]:.3f}")
P
