#!/usr/bin/env python3
"""A few hundred steps of one of the reference's example layouts (PMC_CMD of tools/pmc_passes.sh for the small kernels, e.g.
PMC_FILTER=k_mdbc PMC_CMD='python tools/steps_example.py duckling 4 200' tools/pmc_passes.sh r4_pmc_mdbc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
from sphexample_amd.engine import make_engine  # noqa: E402
name, fb, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
p, s = getattr(conftest, "load_" + name)()
e = make_engine(p, s, device_float_bytes=fb)
if hasattr(p, "geometries"):
    e.set_motions(p.geometries)
e.advance(1e9, max_steps=steps)
