import os, sys, time
ROOT = os.environ.get("PROBE_TREE", "/root/repo")
sys.path.insert(0, ROOT)
origin = os.path.join(ROOT, "tools", "bench_variants.py")
sys.argv = sys.argv[:1] + ["200"]
ns = {"__name__": "bv", "__file__": origin}
exec(compile(open(origin).read().split("\nfor dp in")[0], origin, "exec"), ns)
p, s0 = ns["dam_break_3d"](0.02), ns["setup_dam_break_3d"](0.02)
from sphexample_amd.engine import make_engine
for model in ("laminar", "default"):
    e = make_engine(p, ns["models"](s0, model), device_float_bytes=4)
    e.advance(1e9, max_steps=20)
    for n in (200, 200, 400, 100, 100, 50):
        t0 = time.perf_counter(); pr = e.advance(1e9, max_steps=n); dt = time.perf_counter() - t0
        print(f"{model} advance({n}): {dt * 1e3:7.2f} ms = {dt / n * 1e6:6.1f} us/step  rebuilds so far {pr.n_rebuilds}", flush=True)
    print({k.split()[0]: (round(v[0] * 1e3, 2), v[1]) for k, v in e.timers().items()})
