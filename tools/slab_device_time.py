#!/usr/bin/env python3
"""Device WORK per step of the slabs of bench.py's weak-scaling workloads, free of the contention of N slabs sharing one GPU.

tools/scaling_inputs.py times N slabs side by side on the one chip: their launches overlap and slow each other down, so the step time
per slab over-states what a GPU of its own would need.  Here every launch runs ALONE (`AMD_SERIALIZE_KERNEL=3`: the runtime waits for
each kernel before it starts the next) under `rocprofv3 --kernel-trace`, and the durations are added up per kernel name: the sum over
the kernels of a step, divided by the slabs, is the device time one slab's step needs on a chip of its own — before any overlap of the
halo with the interior launch and without launch gaps.  Two runs per configuration (K1 and K2 steps) and their difference per step take
the upload, the first rebuilds and the warm-up out.

  python tools/slab_device_time.py            (parent: runs itself under rocprofv3 for 1, 2, 4, 8 slabs; JSON on stdout; ≈8 GPU-minutes)
  python tools/slab_device_time.py child <slabs> <steps>
"""
import json
import os
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DP1 = 0.00425
K1, K2 = 30, 90


def child(world, steps, whole=False):
    from sphexample_amd.cases import dam_break_3d, setup_dam_break_3d
    from sphexample_amd.engine import make_engine
    dp = DP1 / world ** (1.0 / 3.0)
    p, s = dam_break_3d(dp), setup_dam_break_3d(dp)
    # (`whole`: the SAME lattice on one plain engine — what the workload itself costs per particle at that resolution, without any slab)
    e = make_engine(p, s, device_float_bytes=4, devices=[0] * world if world > 1 and not whole else None)
    pr = e.advance(1e9, max_steps=steps)
    print(json.dumps({"N": len(p), "steps": int(pr.steps_done), "rebuilds": int(pr.n_rebuilds)}), flush=True)


def kernel_sums(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()
    return {r[0]: (int(r[1]), float(r[2])) for r in rows}


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("sphmi::", "")
    return n if len(n) < 70 else n[:67] + "..."


def profiled(world, steps, whole=False):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", TMPDIR="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "child", str(world), str(steps)] + (["whole"] if whole else [])
        pr = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=1500)
        line = [x for x in pr.stdout.splitlines() if x.startswith("{")]
        if pr.returncode != 0 or not line:
            raise RuntimeError(pr.stderr[-2000:])
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        return json.loads(line[-1]), kernel_sums(dbs[0])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]), int(sys.argv[3]), whole=len(sys.argv) > 4)
    out = {"method": "AMD_SERIALIZE_KERNEL=3 + rocprofv3 --kernel-trace, sum of kernel durations, (K2 - K1)-step difference", "K1": K1, "K2": K2, "runs": []}
    for world in (1, 2, 4, 8):
        (i1, s1), (i2, s2) = profiled(world, K1), profiled(world, K2)
        dk = i2["steps"] - i1["steps"]
        per = {}
        for name in set(s1) | set(s2):
            c = (s2.get(name, (0, 0.0))[0] - s1.get(name, (0, 0.0))[0]) / dk
            t = (s2.get(name, (0, 0.0))[1] - s1.get(name, (0, 0.0))[1]) / dk
            if abs(t) > 50.0:                                   # ns per step
                per[short(name)] = {"launches_per_step": round(c, 2), "us_per_step": round(t / 1e3, 2)}
        total_us = sum(v["us_per_step"] for v in per.values())
        run = {"slabs": world, "N": i2["N"], "rebuilds_between": i2["rebuilds"] - i1["rebuilds"], "device_us_per_step_all_slabs": round(total_us, 1),
               "device_us_per_step_per_slab": round(total_us / world, 1), "us_per_particle_step": total_us / i2["N"],
               "kernels": dict(sorted(per.items(), key=lambda kv: -kv[1]["us_per_step"]))}
        if world > 1:
            (j1, w1), (j2, w2) = profiled(world, K1, True), profiled(world, K2, True)
            whole_us = sum((w2.get(n, (0, 0.0))[1] - w1.get(n, (0, 0.0))[1]) for n in set(w1) | set(w2)) / (j2["steps"] - j1["steps"]) / 1e3
            run["same_lattice_on_one_engine_us_per_step"] = round(whole_us, 1)
            run["same_lattice_on_one_engine_us_per_particle_step"] = whole_us / j2["N"]
            run["slab_work_overhead_vs_same_lattice_on_one_engine"] = run["us_per_particle_step"] / (whole_us / j2["N"]) - 1.0
        out["runs"].append(run)
        print(f"[slab device time] {world} slab(s), N = {i2['N']}: {total_us / world:.1f} us of kernels per step and slab "
              f"({total_us / i2['N'] * 1e3:.4f} ns per particle-step)" + (f"; the same lattice on one engine {run['same_lattice_on_one_engine_us_per_particle_step'] * 1e3:.4f}" if world > 1 else ""),
              file=sys.stderr, flush=True)
    base = out["runs"][0]["us_per_particle_step"]
    for r in out["runs"]:
        r["per_particle_cost_vs_1.06M_one_device"] = r["us_per_particle_step"] / base - 1.0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
