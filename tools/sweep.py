#!/usr/bin/env python3
"""Build libsphmi variants with different -D switches and bench each (run on the GPU box).
usage: python tools/sweep.py "name1:-DSPHMI_KSLOTS=16 -DX=1" "name2:..." [-- bench args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd import build  # noqa: E402

args = sys.argv[1:]
bench_args = ["--steps", "30", "--warmup", "3", "--no-cpu-baseline"]
if "--" in args:
    i = args.index("--")
    bench_args = args[i + 1:]
    args = args[:i]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for spec in args:
    name, _, flags = spec.partition(":")
    out = f"/tmp/libsphmi_{name}.so"
    try:
        build.build(force=True, extra_flags=flags.split(), out=out)
    except Exception as e:
        print(f"{name}: BUILD FAILED {e}")
        continue
    env = dict(os.environ, SPHMI_LIB=out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *bench_args], env=env,
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"{name}: FAILED\n{r.stdout[-500:]}\n{r.stderr[-1500:]}")
        continue
    for l in r.stderr.splitlines():
        if l.startswith("[sphmi"):
            print(f"{name}: {l}")
    j = json.loads(line[-1])
    print(f"{name:28s} flags[{flags}]  {j['value']:.4g} upd/s  step {j['ms_per_step']:.3f} ms  "
          f"force-kernel {j['roofline']['avg_launch_ms']:.3f} ms", flush=True)
