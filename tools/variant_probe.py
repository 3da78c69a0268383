#!/usr/bin/env python3
"""Step time of ONE kernel instantiation of tools/bench_variants.py: python tools/variant_probe.py <dp> <model> <float bytes> [steps]
(with $SPHMI_LIB pointing at an A/B build: which change moved this instantiation?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv, args = sys.argv[:1] + ["1"], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("bv", os.path.join(os.environ.get("PROBE_TREE", ROOT), "tools", "bench_variants.py"))
src = open(spec.origin).read().split("\nfor dp in")[0]          # the helpers only
ns = {"__name__": "bv", "__file__": spec.origin}; exec(compile(src, spec.origin, "exec"), ns)
dp, model, fb = float(args[0]), args[1], int(args[2]); ns["steps"] = int(args[3]) if len(args) > 3 else 400
p, s0 = ns["dam_break_3d"](dp), ns["setup_dam_break_3d"](dp)
print(f"dp {dp} N={len(p)} {model} fp{8 * fb}: {ns['run'](p, ns['models'](s0, model), fb):.1f} us/step   [{os.environ.get('SPHMI_LIB', 'in-tree')}]")
