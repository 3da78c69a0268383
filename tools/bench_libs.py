#!/usr/bin/env python3
"""Bench every prebuilt variant under build/variants/ (tools/prebuild_variants.py), `reps` runs each, interleaved so that
clock drift of the box hits all variants alike.  usage: python tools/bench_libs.py [reps] [name ...] [-- bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
bench_args = ["--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
if "--" in args:
    i = args.index("--"); bench_args = args[i + 1:]; args = args[:i]
reps = int(args[0]) if args and args[0].isdigit() else 2
names = [a for a in args if not a.isdigit()]
vdir = os.path.join(ROOT, "build", "variants")
libs = sorted(f for f in os.listdir(vdir) if f.endswith(".so") and (not names or f[len("libsphmi_"):-3] in names))
res = {l: [] for l in libs}
for r in range(reps):
    for l in libs:
        env = dict(os.environ, SPHMI_LIB=os.path.join(vdir, l))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *bench_args], env=env, capture_output=True, text=True)
        line = [x for x in p.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(f"{l}: FAILED\n{p.stderr[-800:]}", flush=True); continue
        j = json.loads(line[-1])
        res[l].append((j["value"], j["roofline"]["avg_launch_ms"]))
for l in libs:
    if res[l]:
        flags = open(os.path.join(vdir, l + ".flags")).read().strip() if os.path.exists(os.path.join(vdir, l + ".flags")) else ""
        v = [a for a, _ in res[l]]; k = [b for _, b in res[l]]
        print(f"{l[len('libsphmi_'):-3]:24s} best {max(v):.4e}  mean {sum(v)/len(v):.4e} upd/s   kernel best {min(k):.4f} mean {sum(k)/len(k):.4f} ms   [{flags}]", flush=True)
