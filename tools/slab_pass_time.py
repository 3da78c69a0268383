#!/usr/bin/env python3
"""What ONE slab's neighbour pass costs with the chip to itself — interior launch beside unpack + slab-edge launch, halo already landed.

`SPHMI_DD_ONE_SLAB_AT_A_TIME=1` (csrc/sphmi_multi.h, pass_one_at_a_time): the slabs of a one-process handle that share GPU 0 take their
passes one after the other; the host times each slab's pass between two synchronisations (sphmi_multi_halo_info words 10, 11).  For
bench.py's weak-scaling workloads (N = 2, 4, 8 slabs; N = 8 is BASELINE config 4) and both choices of the slab-edge launch's waves per
tile ($SPHMI_EDGE_WPT_JOINT = 0: the edge list chooses for itself; 1: it counts the interior tiles it runs beside), next to the step of
ONE plain engine on the same lattice (its per-particle cost is what the workload itself costs at that resolution).

  python tools/slab_pass_time.py [steps] > gpurun_out/r06/slab_pass_time.json          (≈3 GPU-minutes)
  python tools/slab_pass_time.py child <slabs> <steps>                                  (one configuration; the environment decides the mode)
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DP1 = 0.00425
KEYS = ("slab", "rows", "send_l_A", "send_r_A", "send_l_H", "send_r_H", "tiles_interior", "tiles_edge", "run_interior", "run_edge", "pass1_alone_ns", "pass2_alone_ns")


def child(world, steps):
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    dp = DP1 / world ** (1.0 / 3.0)
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    whole = os.environ.get("SLAB_PASS_WHOLE") == "1"
    e = make_engine(p, s, device_float_bytes=4, devices=None if whole else [0] * world)
    e.advance(1e9, max_steps=20)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); e.advance(1e9, max_steps=steps); best = min(best, time.perf_counter() - t0)
    res = {"slabs": world, "N": len(p), "dp": dp, "whole_lattice_on_one_engine": whole, "ms_per_step": best / steps * 1e3}
    if whole:
        res["force_kernel_avg_launch_ms"] = e.force_kernel_stats()[0]
    else:
        out = (C.c_int64 * (12 * world))()
        n = C.c_int32()
        e._lib.sphmi_multi_halo_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]
        e._check(e._lib.sphmi_multi_halo_info(e._h, out, 12 * world, C.byref(n)))
        res["per_slab"] = [dict(zip(KEYS, out[12 * k:12 * k + 12])) for k in range(n.value // 12)]
        res["owned"] = e.owned_count()
    print(json.dumps(res), flush=True)


def run_child(world, steps, env):
    pr = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(world), str(steps)], env=env, capture_output=True, text=True, timeout=1500)
    line = [x for x in pr.stdout.splitlines() if x.startswith("{")]
    if pr.returncode != 0 or not line:
        raise RuntimeError(pr.stderr[-1500:])
    return json.loads(line[-1])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    out = {"steps": steps, "runs": []}
    for world in (1, 2, 4, 8):
        w = run_child(world, steps, dict(os.environ, SLAB_PASS_WHOLE="1"))
        out["runs"].append(w)
        print(f"[slab pass] the {world}-GPU lattice (N = {w['N']}) on ONE plain engine: {w['ms_per_step'] * 1e3:.1f} us per step = "
              f"{w['ms_per_step'] * 1e6 / w['N']:.4f} ns per particle-step", file=sys.stderr, flush=True)
        if world == 1:
            continue
        for joint in (0, 1):
            r = run_child(world, steps, dict(os.environ, SPHMI_DD_ONE_SLAB_AT_A_TIME="1", SPHMI_EDGE_WPT_JOINT=str(joint)))
            r["edge_wpt_joint"] = joint
            tot = [(q["pass1_alone_ns"] + q["pass2_alone_ns"]) / 1e3 for q in r["per_slab"]]
            r["passes_alone_us_per_step"] = {"mean": sum(tot) / len(tot), "slowest": max(tot), "per_slab": tot}
            r["vs_whole_lattice_per_particle"] = (sum(tot) / r["owned"]) / (w["ms_per_step"] * 1e3 / w["N"]) - 1.0
            out["runs"].append(r)
            print(f"[slab pass] {world} slabs, edge waves per tile joint = {joint}: the two passes of a slab alone on the chip: mean {sum(tot) / len(tot):.1f} us, "
                  f"slowest slab {max(tot):.1f} us per step; summed over the slabs {r['vs_whole_lattice_per_particle'] * 100:+.1f} % per particle against the plain engine",
                  file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
