#!/usr/bin/env python3
"""Run one example case for K steps (profiling target).  usage: python tools/run_example.py <case> [steps] [float bytes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest
from sphexample_amd.engine import make_engine
name, steps, fb = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 4
p, s = getattr(conftest, "load_" + name)()
e = make_engine(p, s, device_float_bytes=fb)
if hasattr(p, "geometries"):
    e.set_motions(p.geometries)
e.advance(1e9, max_steps=steps)
