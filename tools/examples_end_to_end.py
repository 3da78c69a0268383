"""Round 5 (round-4 review, "Next round" 3): every stock example of the reference run END TO END — to its own SimulationTime, through the
per-interval call pattern of `RunSimulation` (/root/reference/src/SPHCellList.jl:881-929: one SimulationLoop call = one `sphmi_advance`
+ one download per OutputTimes interval) — with the record the reference's README quotes a single figure for
(`example/Dambreak3d.jl` at dx = 0.0085, 1.6 s, output every 0.01 s: "1+ day" on the CPU, /root/reference/README.md:11).

For each case and precision: wall seconds of the whole RunSimulation (asynchronous output) and of the same run without downloads (the
share of output), steps, rebuilds by where they ran (sphmi_timers rows 02a-02d), the largest dense cell grid the open domain asked for
against max_cells (2^27), energies and the fp32 - fp64 differences at the end.

    python tools/examples_end_to_end.py [--short] [case ...]        → JSON lines + a markdown table on stdout
"""
import copy
import json
import sys
import time

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np

import conftest
from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
from sphexample_amd.engine import Engine
from sphexample_amd.simulation import RunSimulation

SHORT = "--short" in sys.argv


def load(name):
    if name == "dam_break_3d_dx0.0085":                       # example/Dambreak3d.jl:8 (the input files of that resolution are not in the checkout: generated)
        return dam_break_3d(0.0085), setup_dam_break_3d(0.0085)
    if name == "dam_break_3d_c3":                             # round 6: the headline size (BASELINE config 3, dp = 0.00425, 1.06 M particles), 0 → 0.4 s, output every 0.01 s
        import dataclasses
        s = setup_dam_break_3d(0.00425)
        return dam_break_3d(0.00425), dataclasses.replace(s, SimMetaData=dataclasses.replace(s.SimMetaData, SimulationTime=0.4))
    return getattr(conftest, "load_" + name)()


CASES = ["dam_break_3d_dx0.0085", "dam_break_3d_shipped", "still_wedge", "dam_break_2d_mdbc", "still_wedge_middle_square", "duckling", "moving_square"]


def grid_cells(pos, H):
    c = np.floor(np.abs(pos) / H + 0.5) * np.sign(pos)
    ext = c.max(axis=0) - c.min(axis=0) + 3                    # + one empty layer on either side, as the engine pads
    return int(np.prod(ext))


def run(name, fb, with_output):
    p0, s = load(name)
    p = p0.copy()
    meta = copy.deepcopy(s.SimMetaData)
    if SHORT:
        meta.SimulationTime = (meta.OutputTimes if np.isscalar(meta.OutputTimes) else meta.OutputTimes[0]) * 5
    rec = {"cells_max": 0, "outputs": 0, "timers": None, "fb": None}
    holder = {}

    def factory(cfg):
        if "--oracle" in sys.argv:                              # (dry run of this script on a machine without a GPU: tests/ may use the oracle, tools/ too)
            from oracle.oracle import Oracle
            holder["eng"] = Oracle(cfg)
            holder["eng"].timers = lambda: {}
            holder["eng"].device_float_bytes = 8
        else:
            holder["eng"] = Engine(cfg)
        return holder["eng"]

    def on_output(md, P):
        rec["outputs"] += 1
        if rec["outputs"] % 8 == 1:
            rec["cells_max"] = max(rec["cells_max"], grid_cells(P.Position, s.SimKernel.H))
        rec["timers"] = holder["eng"].timers(); rec["fb"] = holder["eng"].device_float_bytes
        rec["last"] = (md.Iteration, md.TotalTime)

    geo = getattr(p0, "geometries", None)
    t0 = time.perf_counter()
    steps = RunSimulation(SimGeometry=geo, SimMetaData=meta, SimConstants=s.SimConstants, SimKernel=s.SimKernel, SimParticles=p,
                          SimViscosity=s.SimViscosity, SimDensityDiffusion=s.SimDensityDiffusion, device_float_bytes=fb,
                          on_output=on_output if with_output else None, async_output=True, backend_factory=factory)
    wall = time.perf_counter() - t0
    fluid = p.Type == 1
    m0, g = s.SimConstants.m0, s.SimConstants.g
    ke = 0.5 * m0 * float((p.Velocity[fluid] ** 2).sum())
    pe = m0 * g * float(p.Position[fluid, -1].sum())
    out = dict(case=name, fb=fb, N=len(p), wall_s=wall, intervals=len(steps), iteration=int(meta.Iteration), t=float(meta.TotalTime),
               ke=ke, pe=pe, rho_mean=float(p.Density[fluid].mean()), rho_max=float(p.Density.max()), rho_min=float(p.Density.min()),
               nan=int(np.isnan(p.Position).sum()), cells_max=rec["cells_max"], outputs=rec["outputs"])
    if rec["timers"]:
        out["resolved_fb"] = rec["fb"]
        out["rebuilds"] = {k: int(v[1]) for k, v in rec["timers"].items() if k.startswith("02")}
    idx = np.argsort(p.ID)
    out["_state"] = (p.Position[idx].copy(), p.Density[idx].copy())
    return out


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or CASES
    rows = []
    for name in names:
        res = {}
        for fb in (4, 8):
            a = run(name, fb, True)
            b = run(name, fb, False)
            a["wall_no_output_s"] = b["wall_s"]
            res[fb] = a
        d_x = np.abs(res[4]["_state"][0] - res[8]["_state"][0]).max()
        for fb in (4, 8):
            r = res[fb]
            r.pop("_state")
            r["fp32_minus_fp64"] = dict(ke_rel=(res[4]["ke"] - res[8]["ke"]) / max(abs(res[8]["ke"]), 1e-300), pe_rel=(res[4]["pe"] - res[8]["pe"]) / max(abs(res[8]["pe"]), 1e-300),
                                        rho_mean_rel=(res[4]["rho_mean"] - res[8]["rho_mean"]) / res[8]["rho_mean"], x_max_abs=float(d_x))
            print(json.dumps(r), flush=True)
            rows.append(r)
    print("\n| case | N | kernels | sim. time | steps | intervals | wall s (with output) | wall s (no output) | updates/s | rebuilds (host / identity / device / repeated) | max dense grid | ΔKE, ΔPE, Δρ̄ (fp32 − fp64, relative) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        rb = r.get("rebuilds", {})
        rbs = " / ".join(str(rb.get(k, 0)) for k in sorted(rb))
        d = r["fp32_minus_fp64"]
        print(f"| {r['case']} | {r['N']} | fp{8 * r['fb']} | {r['t']:.3f} | {r['iteration']} | {r['intervals']} | {r['wall_s']:.2f} | {r['wall_no_output_s']:.2f} | "
              f"{r['N'] * r['iteration'] / r['wall_no_output_s']:.3g} | {rbs} | {r['cells_max']:,} | {d['ke_rel']:+.1e}, {d['pe_rel']:+.1e}, {d['rho_mean_rel']:+.1e} |")


if __name__ == "__main__":
    main()
