#!/usr/bin/env python3
"""A SEQUENCE of instantiations in one process, as tools/bench_variants.py runs them: python tools/variant_probe_seq.py <steps> <dp:model:fb> ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tree = os.environ.get("PROBE_TREE", ROOT)
sys.path.insert(0, tree)
args = sys.argv[1:]
sys.argv = sys.argv[:1] + [args[0]]
origin = os.path.join(tree, "tools", "bench_variants.py")
ns = {"__name__": "bv", "__file__": origin}
exec(compile(open(origin).read().split("\nfor dp in")[0], origin, "exec"), ns)
for spec in args[1:]:
    dp, model, fb = spec.split(":")
    p, s0 = ns["dam_break_3d"](float(dp)), ns["setup_dam_break_3d"](float(dp))
    print(f"dp {dp} N={len(p)} {model} fp{8 * int(fb)}: {ns['run'](p, ns['models'](s0, model), int(fb)):.1f} us/step", flush=True)
