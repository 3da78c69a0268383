#!/usr/bin/env python3
"""Bench a matrix of prebuilt variants × environment settings, interleaved, `reps` times.
usage: python tools/bench_env_matrix.py reps "lib1,lib2" "ENV1=a ENV2=b" "ENV1=c" ... [-- bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
bench_args = ["--steps", "60", "--warmup", "5", "--no-cpu-baseline", "--no-extras"]
if "--" in args:
    i = args.index("--"); bench_args = args[i + 1:]; args = args[:i]
reps, libs, envs = int(args[0]), args[1].split(","), args[2:] or [""]
vdir = os.path.join(ROOT, "build", "variants")
res = {}
for r in range(reps):
    for l in libs:
        for e in envs:
            env = dict(os.environ, SPHMI_LIB=os.path.join(vdir, f"libsphmi_{l}.so"))
            env.update(kv.split("=", 1) for kv in e.split())
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *bench_args], env=env, capture_output=True, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(f"{l} [{e}]: FAILED\n{p.stderr[-600:]}", flush=True); continue
            j = json.loads(line[-1])
            res.setdefault((l, e), []).append((j["value"], j["roofline"]["avg_launch_ms"]))
for (l, e), v in res.items():
    a = [x for x, _ in v]; k = [y for _, y in v]
    print(f"{l:12s} {e:40s} best {max(a):.4e} mean {sum(a)/len(a):.4e} upd/s  kernel best {min(k):.4f} mean {sum(k)/len(k):.4f} ms", flush=True)
