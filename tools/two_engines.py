#!/usr/bin/env python3
"""Is the slow start of a bench window the hardware or the simulation?  (i) two engines one after the other from the same
initial state; (ii) one engine, warmed up, with pauses between its advance calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from sphexample_amd.cases import setup_dam_break_3d
from sphexample_amd.engine import make_generated_dam_break_engine
dp = 0.00425
def run(e, K, tag):
    e.force_kernel_stats(reset=True)
    t0 = time.perf_counter(); e.advance(1e9, max_steps=K); dt = time.perf_counter() - t0
    ms, n = e.force_kernel_stats()
    print(f"{tag:34s} K={K:3d}  {dt / K * 1e3:.3f} ms/step  kernel {ms:.4f} ms ({n} sampled launches)", flush=True)
e = make_generated_dam_break_engine(dp, setup_dam_break_3d(dp), device_float_bytes=4)
for K in (5, 20, 20, 20):
    run(e, K, "fresh engine")
for pause in (1.0, 0.0, 0.2, 0.0, 0.05, 0.0):
    time.sleep(pause)
    run(e, 20, f"same engine after {pause:.2f} s idle")
