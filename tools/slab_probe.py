import os, sys, time
tree = sys.argv[1]; dp = float(sys.argv[2]); model = sys.argv[3]; nd = int(sys.argv[4])
sys.path.insert(0, tree); sys.argv = sys.argv[:1] + ["1"]
src = open(os.path.join(tree, "tools", "bench_variants.py")).read().split("\nfor dp in")[0]
ns = {"__name__": "bv", "__file__": os.path.join(tree, "tools", "bench_variants.py")}; exec(compile(src, "bv", "exec"), ns)
ns["steps"] = 300
p, s0 = ns["dam_break_3d"](dp), ns["setup_dam_break_3d"](dp)
print(f"{os.path.basename(tree.rstrip('/')):8s} dp {dp} {model} {nd} slabs: {ns['run'](p, ns['models'](s0, model), 4, devices=[0] * nd):.1f} us/step", flush=True)
