"""fp32 handles against the fp64 oracle over >= 100 steps on the example layouts the Julia shim runs in fp32 by default
(round-4 review, "Next round" 1): StillWedgeMDBC, Dambreak2dMDBC, StillWedgeMiddleSquareMDBC, DucklingMDBC, MovingSquare2d.
Prints rho / x / v errors (relative to the field maximum), the rebuild count and the clock at every checkpoint, for the
layout as shipped ("rest") and for a streaming state ("flow": the fluid moves fast enough for Δx-triggered rebuilds).

    python tools/fp32_examples_parity.py [case ...] [--fb 4|8] [--steps 50,50,100]
"""
import sys

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np

import conftest
from conftest import flowing
from oracle.oracle import make_oracle
from sphexample_amd.engine import make_engine

CASES = ["still_wedge", "dam_break_2d_mdbc", "still_wedge_middle_square", "duckling", "moving_square"]


def by_id(st):
    order = np.argsort(st["ID"], kind="stable")
    return {k: v[order] for k, v in st.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fb = 4
    chunks = [50, 50, 100]
    states = ["rest", "flow"]
    fl = dict(seed=5, shear=1.0, base=0.8, noise=0.02, rho_scale=0.3)
    for i, a in enumerate(sys.argv):
        if a == "--fb": fb = int(sys.argv[i + 1]); args = [x for x in args if x != sys.argv[i + 1]]
        if a == "--steps": chunks = [int(x) for x in sys.argv[i + 1].split(",")]; args = [x for x in args if x != sys.argv[i + 1]]
        if a == "--flow":
            b, sh, nz, rs = [float(x) for x in sys.argv[i + 1].split(",")]; fl.update(base=b, shear=sh, noise=nz, rho_scale=rs); args = [x for x in args if x != sys.argv[i + 1]]
        if a == "--states": states = sys.argv[i + 1].split(","); args = [x for x in args if x != sys.argv[i + 1]]
    for name in (args or CASES):
        p0, s = getattr(conftest, "load_" + name)()
        for state in states:
            p = p0
            if state == "flow":
                if name == "moving_square":
                    continue                                  # the body itself forces the rebuilds
                p = flowing(p0, **fl)
                if hasattr(p0, "geometries"): p.geometries = p0.geometries
            eng, orc = make_engine(p, s, device_float_bytes=fb), make_oracle(p, s, threads=8)
            if hasattr(p0, "geometries"):
                eng.set_motions(p0.geometries); orc.set_motions(p0.geometries)
            done = 0
            for n in chunks:
                pe, po = eng.advance(1e9, max_steps=n), orc.advance(1e9, max_steps=n)
                done += n
                e, o = by_id(eng.download()), by_id(orc.download())
                rho = np.abs(e["Density"] - o["Density"]) / np.abs(o["Density"]).max()
                x = np.abs(e["Position"] - o["Position"]).max(axis=1) / np.abs(o["Position"]).max()
                v = np.abs(e["Velocity"] - o["Velocity"]).max() / max(np.abs(o["Velocity"]).max(), 1e-12)
                isf = o["Type"] == 1
                print(f"{name:28s} {state:4s} fb{fb} step {done:4d}: rho {rho.max():.2e} (fluid {rho[isf].max():.2e}, bnd {rho[~isf].max():.2e}, n>1e-5 {(rho > 1e-5).sum()})"
                      f" x {x.max():.2e} v {v:.2e} | rebuilds {pe.n_rebuilds}/{po.n_rebuilds} t {pe.total_time:.9e}/{po.total_time:.9e}"
                      f" dt_rel {abs(pe.last_dt - po.last_dt) / po.last_dt:.1e} vmax {np.abs(o['Velocity']).max():.2f}", flush=True)


if __name__ == "__main__":
    main()
