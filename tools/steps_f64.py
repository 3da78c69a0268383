#!/usr/bin/env python3
"""A few steps of an fp64 handle on the bench lattice (PMC_CMD of tools/pmc_passes.sh for the fp64 kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import make_engine
dp = 0.00425
e = make_engine(dam_break_3d(dp), setup_dam_break_3d(dp), device_float_bytes=8)
e.advance(1e9, max_steps=2)
e.advance(1e9, max_steps=6)
