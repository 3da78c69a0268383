#!/usr/bin/env python3
"""Step time over the space of kernel instantiations: 3-D dam-break lattices of several sizes × float type × physics models
(compiled-in default / run-time models) × slabs in one handle.  Uses only the API round 2 already had, so that the same file
runs inside an older tree (e.g. `build/r2tree`, a `git archive` of the round-2 commit with its library built in place) and the
two tables can be compared line by line: a kernel variant that got slower shows here, not in the parity suite.

  python tools/bench_variants.py [steps]            (≈1.5 GPU-minutes)
"""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sphexample_amd import (ArtificialViscosity, ComplexDensityDiffusion, CubicSpline, Laminar, LaminarSPS,  # noqa: E402
                            LinearDensityDiffusion, PlanarShifting, SPHKernelInstance, StoreKernelOutput, WendlandC2,
                            ZeroDensityDiffusion, ZeroGravityLinearDensityDiffusion, ZeroViscosity)
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d  # noqa: E402
from sphexample_amd.engine import make_engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def models(s, name):
    meta, kern, visc, ddt = s.SimMetaData, s.SimKernel, s.SimViscosity, s.SimDensityDiffusion
    if name == "default":
        pass
    elif name == "laminar":
        visc = Laminar()
    elif name == "sps+complex":
        visc, ddt = LaminarSPS(), ComplexDensityDiffusion()
    elif name == "zero+zero":
        visc, ddt = ZeroViscosity(), ZeroDensityDiffusion()
    elif name == "zerograv":
        ddt = ZeroGravityLinearDensityDiffusion()
    elif name == "shifting":
        meta = dataclasses.replace(meta, SMode=PlanarShifting)
    elif name == "kernel-output":
        meta = dataclasses.replace(meta, KMode=StoreKernelOutput)
    elif name == "cubic":
        kern = SPHKernelInstance(3, CubicSpline(0.2), h=kern.h, k=kern.k)
    elif name == "k1.5":
        kern = SPHKernelInstance(3, WendlandC2(), h=kern.h, k=1.5)
    return dataclasses.replace(s, SimKernel=kern, SimMetaData=meta, SimViscosity=visc, SimDensityDiffusion=ddt)


def run(p, s, fb, devices=None):
    e = make_engine(p, s, device_float_bytes=fb, devices=devices)
    e.advance(1e9, max_steps=20)
    best = 1e9
    # (best of FOUR since round 5: a handle's first 30-40 ms run on a device whose clock governor is still leaving the idle state the upload put it in — with
    # best-of-two a 17 k-particle handle whose first call got FASTER (16 instead of 23 ms) showed up 14 % slower, its second call still inside the ramp)
    for _ in range(4):
        t0 = time.perf_counter(); e.advance(1e9, max_steps=steps); best = min(best, time.perf_counter() - t0)
    return best / steps * 1e6


for dp in (0.02, 0.0115, 0.0085, 0.0057, 0.00425):
    p, s0 = dam_break_3d(dp), setup_dam_break_3d(dp)
    names = ("default", "laminar", "sps+complex", "zero+zero", "zerograv", "shifting", "kernel-output", "cubic", "k1.5")
    if dp == 0.00425:
        names = ("default", "laminar", "shifting")
    for name in names:
        s = models(s0, name)
        for fb in (4, 8):
            if dp <= 0.0057 and fb == 8 and name not in ("default", "laminar"):
                continue
            us = run(p, s, fb)
            print(f"dp {dp:<8} N={len(p):8d} {name:14s} fp{fb * 8}  one slab   {us:9.1f} us/step  {len(p) / us * 1e6:.3g} upd/s", flush=True)
    for nd in (2, 4):
        if dp in (0.0115, 0.0057):
            continue
        for name in ("default", "laminar"):
            us = run(p, models(s0, name), 4, devices=[0] * nd)
            print(f"dp {dp:<8} N={len(p):8d} {name:14s} fp32  {nd} slabs    {us:9.1f} us/step  {len(p) / us * 1e6:.3g} upd/s", flush=True)
